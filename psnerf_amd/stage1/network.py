"""Stage-1 occupancy + appearance field with the reference's ``NeuralNetwork`` interface
(stage1/model/network.py:7-138) on the HIP kernels.

state_dict keys are the reference's (``lin{l}.{bias,weight_g,weight_v}``, ``lina{l}.*``), so released
checkpoints load unchanged.  Mapping to kernels:
  * training / with-graph evaluation  -> ops.GeoFieldFused (value pass, reverse-mode spatial-gradient sweep and the
    hand-written adjoints of both as four launches of the register-resident chain engine + one grouped weight-gradient
    launch) and ops.AppNetFused (appearance net); networks the engine does not hold (hidden width != 256) fall back to
    ops.GeoField / ops.ReluMLP (fp32-MFMA GEMM sequences);
  * no-grad occupancy queries (ray-march sweep, shadow rays) -> the lean register-resident engine
    (fused.pack_geo_occupancy), re-packed once per optimiser step; the secant refinement -> psn_root_find, the same
    engine iterating inside one launch.
Weight normalisation (w = g * v / |v|) of all layers of a network is one launch forward / one backward
(ops.WeightNormAll), so autograd carries dW back to weight_g / weight_v.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import fused, hip, ops


class WNLinear(nn.Module):
    """Parameters of nn.utils.weight_norm(nn.Linear(...)): weight_g [out,1], weight_v [out,in], bias."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight_g = nn.Parameter(weight.norm(2, dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(weight.clone())
        self.bias = nn.Parameter(bias.clone())

    def weight(self):
        v = self.weight_v
        return v * (self.weight_g / v.norm(2, dim=1, keepdim=True))


class NeuralNetwork(nn.Module):
    def __init__(self, cfg_all, **kwargs):
        super().__init__()
        cfg = cfg_all['model']
        hidden = cfg['hidden_dim']
        self.octaves_pe = cfg['octaves_pe']
        self.octaves_pe_views = cfg['octaves_pe_views']
        self.skips = list(cfg['skips'])
        self.rescale = cfg['rescale']
        self.feat_size = cfg['feat_size']
        geometric_init = cfg['geometric_init']
        bias0 = 0.6

        d_pe = 3 + 6 * self.octaves_pe
        self.d_pe = d_pe
        self.d_view = 3 + 6 * self.octaves_pe_views
        d_app = 3 + self.d_view + 3 + self.feat_size
        dims = [d_pe] + [hidden] * cfg['num_layers'] + [self.feat_size + 1]
        self.num_layers = len(dims)
        self.n_geo = len(dims) - 1
        for l in range(self.n_geo):
            d_out = dims[l + 1] - dims[0] if (l + 1) in self.skips else dims[l + 1]
            lin = nn.Linear(dims[l], d_out)  # consumes the RNG like the reference (network.py:45)
            if geometric_init:  # sphere initialisation, network.py:47-61
                with torch.no_grad():
                    if l == self.n_geo - 1:
                        nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                        nn.init.constant_(lin.bias, -bias0)
                    elif self.octaves_pe > 0 and l == 0:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.constant_(lin.weight[:, 3:], 0.0)
                        nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(d_out))
                    elif self.octaves_pe > 0 and l in self.skips:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(d_out))
                        nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                    else:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(d_out))
            setattr(self, 'lin%d' % l, WNLinear(lin.weight.detach(), lin.bias.detach()))
        dims_app = [d_app] + [hidden] * 4 + [3]
        self.num_layers_app = len(dims_app)
        self.n_app = len(dims_app) - 1
        for l in range(self.n_app):
            lin = nn.Linear(dims_app[l], dims_app[l + 1])
            setattr(self, 'lina%d' % l, WNLinear(lin.weight.detach(), lin.bias.detach()))
        self._pack_epoch = 0
        self._packed = None
        self._packed_key = None
        # 'fp32' (default, exact) or 'bf16x6': GRADIENT-FREE occupancy queries (shadow rays, ray-march sweep, shape_extract;
        # stage1/model/rendering.py:297-523) on the split-bf16 engine -- fp32-class arithmetic on the bf16 matrix pipe
        # (csrc/mlp_infer_x3.hip, OCC variant).  Opt-in experiment, never the headline; every training path stays exact fp32.
        self.inference_precision = 'fp32'
        self._packed_x3 = None
        self._packed_x3_key = None
        self._packed_b3 = None
        self._packed_b3_key = None
        self._chains = None
        self._chains_key = None
        self._app_packed = None
        self._app_key = None

    # ---- weight-pack caches ---------------------------------------------------------------------
    def _params_key(self):
        """Version counter and storage address of EVERY parameter + the invalidation epoch (see invalidate_packs)."""
        return (self._pack_epoch,) + tuple((int(q._version), q.data_ptr()) for q in self.parameters())

    def invalidate_packs(self):
        """Drop the cached occupancy / chain packs.  Needed only after editing parameters through ``.data`` (EMA swaps,
        manual clipping): such edits bump neither the version counters nor the storage addresses the caches key on."""
        self._pack_epoch += 1

    def _apply(self, fn, *a, **k):
        self._pack_epoch += 1
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pack_epoch += 1
        return super()._load_from_state_dict(*a, **k)

    # ---- effective weights ----------------------------------------------------------------------
    def _effective(self, prefix, n, scales):
        """Effective matrices of layers ``prefix``0..n-1.  On the GPU all of them come from ONE launch
        (ops.WeightNormAll, differentiable); elsewhere (state-dict handling on the host) from the torch formula."""
        lins = [getattr(self, '%s%d' % (prefix, l)) for l in range(n)]
        if not lins[0].weight_v.is_cuda:
            return [lin.weight() * s if s != 1.0 else lin.weight() for lin, s in zip(lins, scales)]
        gv = []
        for lin in lins:
            gv += [lin.weight_g, lin.weight_v]
        return list(ops.WeightNormAll.apply(tuple(scales), *gv))

    def _geo_params(self):
        inv = float(1.0 / np.sqrt(2))  # folds the cat[x, pe]/sqrt(2) of network.py:90-91 into the skip layer
        Ws = self._effective('lin', self.n_geo, [inv if l in self.skips else 1.0 for l in range(self.n_geo)])
        out = []
        for l in range(self.n_geo):
            out += [Ws[l], getattr(self, 'lin%d' % l).bias]
        return out

    def _app_params(self):
        Ws = self._effective('lina', self.n_app, [1.0] * self.n_app)
        return Ws, [getattr(self, 'lina%d' % l).bias for l in range(self.n_app)]

    USE_FUSED_CHAINS = True
    MAX_ROWS = 1 << 20  # rows per GeoField call: bounds the saved activations to ~50 GB of the 288 GB HBM

    def _geo_chains(self, params):
        key = (self._params_key(), ops.CHAIN_X3, ops.GEO_SINGLE_DUMP)  # (both are run-time switches: a toggle rebuilds the packs)
        if self._chains is None or self._chains_key != key:
            with torch.no_grad():
                self._chains = fused.pack_geo_chains(params[0::2], params[1::2], self.skips, self.d_pe, single_dump=ops.GEO_SINGLE_DUMP,
                                                     x3=ops.CHAIN_X3)
            self._chains_key = key
        return self._chains

    def _geo_call(self, p_flat, with_grad, params, chains, feat_rows=None):
        if chains is not None:
            return ops.GeoFieldFused.apply(p_flat, self.octaves_pe, 1.0 / self.rescale, tuple(self.skips), with_grad,
                                           chains, feat_rows, *params)
        assert feat_rows is None
        if self.USE_FUSED_CHAINS:  # (False: the layer-wise cross-check of the tests)
            ops.fallback('stage1 geometry network -> layer-wise GEMMs (ops.GeoField)', p_flat,
                         'hidden widths / skips / feature size outside the fused chains')
        o, g = ops.GeoField.apply(p_flat, self.octaves_pe, 1.0 / self.rescale, tuple(self.skips), with_grad, *params)
        return o[:, :1], o[:, 1:], g

    @torch.no_grad()
    def prepack(self, occupancy=False, chains=False):
        """Build the weight packs that depend on nothing but the parameters, ahead of their first use: the occupancy
        pack right after an optimiser step (queued behind the backward pass), the chain packs while the host waits for
        the ray-march sweep.  All of them are cached by parameter version, so the later calls find them ready."""
        if not self.lin0.weight_v.is_cuda:
            return
        if occupancy and self._hidden_is_256():
            self._occupancy_packed()
        if chains and self.USE_FUSED_CHAINS:
            if self._hidden_is_256() and len(self.skips) == 1 and self.feat_size == 256 and self.n_geo <= 10 and self.d_pe <= 64:
                self._geo_chains(self._geo_params())
            Ws, bs = self._app_params()
            d_x = 3 + self.d_view + 3
            if (self.feat_size == 256 and d_x <= 64 and Ws[0].shape[0] == 256
                    and all(w.shape == (256, 256) for w in Ws[1:-1]) and self.n_app <= 10):
                self._app_chains(Ws, bs, d_x)

    def _geo_parts(self, p_flat, with_grad):
        """(logit [Q,1], features [Q,F], d logit / d p [Q,3]) of the geometry network.  256-wide networks run as
        fused register-resident chains (ops.GeoFieldFused); other widths fall back to the GEMM sequence."""
        params = self._geo_params()
        chains = None
        if self.USE_FUSED_CHAINS and self._hidden_is_256() and len(self.skips) == 1 and self.feat_size == 256 \
                and self.n_geo <= 10 and self.d_pe <= 64:
            chains = self._geo_chains(params)
        if p_flat.shape[0] <= self.MAX_ROWS:
            return self._geo_call(p_flat, with_grad, params, chains)
        parts = [self._geo_call(p_flat[s:s + self.MAX_ROWS], with_grad, params, chains)
                 for s in range(0, p_flat.shape[0], self.MAX_ROWS)]
        return tuple(torch.cat([q[i] for q in parts], 0) for i in range(3))

    def _geo(self, p_flat, with_grad):
        logit, feat, grad = self._geo_parts(p_flat, with_grad)
        return torch.cat([logit, feat], dim=1), grad

    def _occupancy_packed(self, allow_x3=False):
        """The packed occupancy network of the exact-fp32 engine; ``allow_x3``: callers that only use ``.on_points`` get the
        split-bf16 pack instead when the module's ``inference_precision`` is 'bf16x6' (opt-in, gradient-free queries)."""
        prec = getattr(self, 'inference_precision', 'fp32')
        if allow_x3 and prec == 'bf16x6' and not torch.is_grad_enabled():
            return self._occupancy_packed_x3()
        if allow_x3 and prec == 'bf16x3' and not torch.is_grad_enabled():
            # the exact engine's kernels on split-bf16 weight stages (two bf16 pieces per operand, three partial products:
            # PsnMlpDesc.w_format = PSN_W_BF16X2; ~1e-5 relative) -- same call surface, the sweep included
            key = self._params_key()
            if getattr(self, '_packed_b3', None) is None or self._packed_b3_key != key:
                with torch.no_grad():
                    Ws = self._effective('lin', self.n_geo, [1.0] * self.n_geo)
                    bs = [getattr(self, 'lin%d' % l).bias for l in range(self.n_geo)]
                    self._packed_b3 = fused.pack_geo_occupancy(Ws, bs, self.skips, self.d_pe, x3=True)
                self._packed_b3_key = key
            return self._packed_b3
        key = self._params_key()
        if self._packed is None or self._packed_key != key:
            with torch.no_grad():
                # without the 1/sqrt(2) fold: pack_geo_occupancy applies it itself
                Ws = self._effective('lin', self.n_geo, [1.0] * self.n_geo)
                bs = [getattr(self, 'lin%d' % l).bias for l in range(self.n_geo)]
                self._packed = fused.pack_geo_occupancy(Ws, bs, self.skips, self.d_pe)
            self._packed_key = key
        return self._packed

    def _occupancy_packed_x3(self):
        """The occupancy network for the split-bf16 engine (opt-in: ``inference_precision = 'bf16x6'``; gradient-free queries
        only -- shadow rays, ray march, shape_extract; rendering.py:378-523): fp32-class arithmetic on the bf16 matrix pipe."""
        key = self._params_key()
        if getattr(self, '_packed_x3', None) is None or self._packed_x3_key != key:
            with torch.no_grad():
                Ws = self._effective('lin', self.n_geo, [1.0] * self.n_geo)
                bs = [getattr(self, 'lin%d' % l).bias for l in range(self.n_geo)]
                self._packed_x3 = fused.pack_geo_occupancy_x3(Ws, bs, self.skips, self.d_pe)
            self._packed_x3_key = key
        return self._packed_x3

    # ---- reference API ----------------------------------------------------------------------------
    def infer_occ(self, p):
        shp = p.shape[:-1]
        out, _ = self._geo(p.reshape(-1, 3), False)
        return out.reshape(*shp, -1)

    def gradient(self, p, tflag=True):
        """d occ_logit / d p -> [Q,1,3] (network.py:108-120).  tflag=False returns a detached result."""
        flat = p.reshape(-1, 3)
        if tflag:
            _, g = self._geo(flat, True)
        else:
            with torch.no_grad():
                _, g = self._geo(flat, True)
        return g.unsqueeze(1)

    def infer_app(self, points, normals, view_dirs, feature_vectors):
        x = torch.cat([points, view_dirs, normals.squeeze(-2), feature_vectors], dim=-1)
        return self._app(x)

    def _app_chains(self, Ws, bs, d_x):
        key = (self._params_key(), ops.CHAIN_X3, ops.GEO_SINGLE_DUMP)  # (both are run-time switches: a toggle rebuilds the packs)
        if self._app_packed is None or self._app_key != key:
            with torch.no_grad():
                self._app_packed = fused.pack_app_chains(Ws, bs, d_x, x3=ops.CHAIN_X3)
            self._app_key = key
        return self._app_packed

    def _app_parts(self, points, view, normal, feat):
        """Colour from (point, raw view direction, normal, geometry features) = infer_app (network.py:128-138) on the
        normalised, encoded view direction without the 289-wide concat: 256-wide networks run as fused chains
        (ops.AppNetFused) whose 64-column input table [p | gamma(v / |v|) | n] is written by one launch (psn_app_input)."""
        d_x = 3 + self.d_view + 3
        Ws, bs = self._app_params()
        if not (self.USE_FUSED_CHAINS and self.feat_size == 256 and d_x <= 64 and Ws[0].shape[0] == 256
                and all(w.shape == (256, 256) for w in Ws[1:-1]) and self.n_app <= 10 and points.is_cuda):
            if self.USE_FUSED_CHAINS:
                ops.fallback('stage1 appearance network -> layer-wise GEMMs (ops.ReluMLP)', points, 'widths / input size outside the fused chains')
            v = view / torch.norm(view, dim=-1, keepdim=True)
            v_pe = ops.positional_encoding(v, self.octaves_pe_views)
            return self._app(torch.cat([points, v_pe, normal, feat], dim=-1))
        x = hip.app_input(points.detach(), view.detach(), normal.detach(), self.octaves_pe_views)
        params = []
        for W, b in zip(Ws, bs):
            params += [W, b]
        y = ops.AppNetFused.apply(x, normal, feat, d_x, self._app_chains(Ws, bs, d_x), *params)
        return torch.tanh(y) * 0.5 + 0.5

    def _app(self, x):
        d = x.shape[-1]
        kp = (d + 3) // 4 * 4
        xp = torch.nn.functional.pad(x.reshape(-1, d), (0, kp - d)).contiguous()
        Ws, bs = self._app_params()
        cols = torch.arange(d, device=x.device)
        y = ops.relu_mlp(xp, cols, -100, False, Ws, bs)
        return (torch.tanh(y) * 0.5 + 0.5).reshape(*x.shape[:-1], 3)

    def occupancy(self, p_flat):
        """sigmoid(-10 * logit) for [Q,3] points without a graph: fused register-resident kernel, positional encoding in
        its prologue (network.py:141-150 + 85-101 in one launch; no [Q,64] table in HBM)."""
        packed = self._occupancy_packed(allow_x3=True)
        return packed.on_points(p_flat.contiguous(), self.octaves_pe, 1.0 / self.rescale)

    def forward(self, p, ray_d=None, only_occupancy=False, return_logits=False, return_addocc=False, noise=False,
                **kwargs):
        shp = p.shape[:-1]
        flat = p.reshape(-1, 3)
        if only_occupancy:
            if not torch.is_grad_enabled() and self.feat_size + 1 > 1 and self._hidden_is_256():
                return self.occupancy(flat).reshape(*shp, 1)
            if not torch.is_grad_enabled():
                ops.fallback('stage1 occupancy queries -> geometry-network formulation', flat, 'hidden width is not 256')
            logit, _, _ = self._geo_parts(flat, False)
            return torch.sigmoid(logit * -10.0).reshape(*shp, 1)
        if ray_d is not None:
            logit, feat, grad = self._geo_parts(flat, True)
            rgb = self._app_parts(flat, ray_d.reshape(-1, 3), grad, feat).reshape(*shp, 3)
            if return_addocc:
                return rgb, torch.sigmoid(logit * -10.0).reshape(*shp, 1)
            return rgb
        if return_logits:
            logit, _, _ = self._geo_parts(flat, False)
            return (-1 * logit).reshape(*shp, 1)
        return None

    def render_and_gradient(self, p, ray_d, extra):
        """``forward(p, ray_d, return_addocc=True)`` and ``gradient(extra)`` from ONE set of geometry-network launches:
        the ``extra`` points (the surface-normal points of rendering.py:200-212, 2 N next to N S render samples) ride
        behind the render samples as rows that are evaluated for d logit / d p only.  Same arithmetic per row as the two
        separate calls; saves four latency-bound chain launches, a weight-gradient launch and the second gradient
        accumulation of every geometry parameter.  -> (rgb [..., 3], occupancy [..., 1], gradient [Qx, 1, 3])."""
        shp = p.shape[:-1]
        flat, ex = p.reshape(-1, 3), extra.reshape(-1, 3)
        q1 = flat.shape[0]
        params = self._geo_params()
        fused_ok = (self.USE_FUSED_CHAINS and self._hidden_is_256() and len(self.skips) == 1 and self.feat_size == 256
                    and self.n_geo <= 10 and self.d_pe <= 64 and q1 + ex.shape[0] <= self.MAX_ROWS and q1 > 0 and flat.is_cuda)
        if not fused_ok:
            rgb, occ = self.forward(p, ray_d, return_addocc=True)
            return rgb, occ, self.gradient(extra)
        logit, feat, grad = self._geo_call(torch.cat([flat, ex], dim=0), True, params, self._geo_chains(params), feat_rows=q1)
        grad_r, grad_x = ops.SplitRows.apply(grad, q1)
        logit_r, _ = ops.SplitRows.apply(logit, q1)
        rgb = self._app_parts(flat, ray_d.reshape(-1, 3), grad_r, feat).reshape(*shp, 3)
        return rgb, torch.sigmoid(logit_r * -10.0).reshape(*shp, 1), grad_x.unsqueeze(1)

    def _hidden_is_256(self):
        return self.lin1.weight_v.shape[1] == 256 and all(
            getattr(self, 'lin%d' % l).weight_v.shape[0] in (256, 256 - self.d_pe) for l in range(self.n_geo - 1))
