"""Stage-2 model: SVBRDF / normal / visibility MLPs + multi-light shading, with the reference's
``PSNetwork`` interface (stage2/model/renderer.py:52-266) on top of the HIP kernels.

Drop-in contract: same constructor (a conf tree with get_string/get_int/get_float/
get_bool), same ``forward(input, albedo_new=None, basis_new=None)`` input/output dictionaries, same
state_dict keys ({albedo,rough,normal,visibility}_net.linears.{i}.{weight,bias}, sgbasis.lobe).

MI355X mapping:
  * positional encodings -> psn_pe_encode tables (one row per surface point / per light), padded to 64
    floats so every row is one aligned MFMA k-tile pair;
  * visibility net: the L shading-light rows (97 % of the FLOPs; no gradient reaches them: renderer.py:197
    detach + loss.py:82-83) and the V supervision-light rows in ONE launch of the register-resident engine
    (ops.VisibilityPair: input block through per-point / per-light init tables, the supervised rows leave their
    activations behind for a chain backward + grouped weight gradients); ops.FusedPairMLP when only the
    shading rows exist (evaluation), the bf16 engine when the caller opts in;
  * albedo / SG-weight / normal nets (128- and 64-wide) -> ops.FusedReluNet (one forward launch with dumps,
    one backward chain, one grouped weight-gradient launch); other widths -> ops.ReluMLP (layer-wise GEMMs);
  * SG / GGX shading -> ops.sg_shade / ops.mf_shade (fused forward / backward kernels);
  * the dense [B, N, C] output dictionary -> one psn_scatter_rows launch (ops.ScatterRows).
"""
import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import fused, hip, ops

PE_STRIDE = 64  # 3 + 6*10 = 63 real columns + 1 zero pad


def params_key(params, epoch=0):
    """Cache key of a weight pack: version counter AND storage address of EVERY parameter (an assign-style
    load_state_dict replaces storages of individual layers), plus the module's invalidation epoch.  In-place edits
    through ``.data`` bump neither: callers that do that (EMA swaps, manual clipping) call ``invalidate_packs()``."""
    return (epoch,) + tuple((int(p._version), p.data_ptr()) for p in params)


def camera_rays(uv, pose, intrinsics):
    """stage2/utils/rend_util.py:90-147 (4x4 pose case), device-agnostic."""
    fx, fy = intrinsics[:, 0, 0], intrinsics[:, 1, 1]
    cx, cy = intrinsics[:, 0, 2], intrinsics[:, 1, 2]
    z = torch.ones_like(uv[:, :, 0])
    x = (uv[:, :, 0] - cx.unsqueeze(-1)) / fx.unsqueeze(-1) * z
    y = (uv[:, :, 1] - cy.unsqueeze(-1)) / fy.unsqueeze(-1) * z
    # d_i = R_i0 x + R_i1 y + R_i2 z as broadcast products (an einsum / bmm here would be the only library GEMM call of the
    # whole path: three multiply-adds per ray need no matrix kernel)
    R = pose[:, :3, :3]
    d = x.unsqueeze(-1) * R[:, None, :, 0] + y.unsqueeze(-1) * R[:, None, :, 1] + z.unsqueeze(-1) * R[:, None, :, 2]
    return F.normalize(d, dim=2), pose[:, :3, 3]


class MLP(nn.Module):
    """Network / Normal_Network (renderer.py:17-49).  Parameters live in ``linears`` exactly like the
    reference; the forward runs on the HIP GEMM path."""

    def __init__(self, din, dout, W, depth, skip_at=(), final='linear'):
        super().__init__()
        self.linears = nn.ModuleList(
            [nn.Linear(din, W)] +
            [nn.Linear(W + din if i in skip_at else W, W) for i in range(depth - 1)] +
            [nn.Linear(W, dout)])
        self.skip_at = list(skip_at)
        self.final = final
        self.din, self.width = din, W

    def invalidate_packs(self):
        """Drop the cached weight packs (call after editing parameters through ``.data``)."""
        self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1

    def _apply(self, fn, *a, **k):
        self.invalidate_packs()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packs()
        return super()._load_from_state_dict(*a, **k)

    def _skip_index(self):
        s = [i for i in self.skip_at if 0 <= i < len(self.linears) - 1]
        assert len(s) <= 1, 'at most one skip connection is supported'
        return s[0] if s else -100

    def weights(self):
        return [l.weight for l in self.linears], [l.bias for l in self.linears]

    FUSED = True  # 64 / 128 / 256-wide networks on encoded points run in the register-resident kernel (ops.FusedReluNet)

    def _fusable(self, x_padded, in_cols):
        Ws = [l.weight for l in self.linears]
        return (self.FUSED and self.width in (64, 128, 256) and x_padded.shape[1] == 64 and not x_padded.requires_grad
                and in_cols.numel() == self.din <= 64 and Ws[-1].shape[0] <= 32 and len(Ws) <= 11
                and all(w.shape[0] == self.width for w in Ws[:-1]))

    def prepack(self):
        """The forward weight pack, rebuilt only when the parameters changed since the last call (evaluation loops and
        the two evaluations of a training step share it)."""
        if not (self.FUSED and self.width in (64, 128, 256) and self.din <= 64 and self.linears[-1].weight.shape[0] <= 32
                and all(l.weight.shape[0] == self.width for l in list(self.linears)[:-1]) and self.linears[0].weight.is_cuda):
            return None
        key = params_key(self.parameters(), getattr(self, '_pack_epoch', 0))
        if getattr(self, '_pack_key', None) != key:
            Ws, bs = self.weights()
            self._pack = ops.FusedReluNet.pack(Ws, bs, self.din, self._skip_index(), self.final == 'sigmoid', self.width)
            self._pack_key = key
        return self._pack

    def forward(self, x_padded, in_cols):
        Ws, bs = self.weights()
        if self._fusable(x_padded, in_cols):
            params = []
            for W, b in zip(Ws, bs):
                params += [W, b]
            return ops.FusedReluNet.apply(x_padded, self.din, self._skip_index(), self.final == 'sigmoid', self.width,
                                          self.prepack(), *params)
        if self.FUSED:  # (FUSED = False is a caller's deliberate choice: the layer-wise cross-check of the tests)
            ops.fallback('stage2.MLP -> layer-wise GEMMs', x_padded, 'width %d, %d layers, input %s' % (self.width, len(Ws), tuple(x_padded.shape)))
        return ops.relu_mlp(x_padded, in_cols, self._skip_index(), self.final == 'sigmoid', Ws, bs)


class SGBasis(nn.Module):
    """stage2/model/sgbasis.py:7-32 (parameter container; evaluation is fused in ops.sg_shade)."""

    def __init__(self, nbasis=9, specular_rgb=False):
        super().__init__()
        self.nbasis, self.specular_rgb = nbasis, specular_rgb
        self.lobe = nn.Parameter(torch.tensor([np.exp(i) for i in range(2, 11)], dtype=torch.float32),
                                 requires_grad=False)


class _Dense(object):
    """A dense output [B, N, C] of PSNetwork.forward that has not been written yet (see forward)."""

    def __init__(self, fill, B, C, rows=None):
        self.fill, self.B, self.C, self.rows = float(fill), int(B), int(C), rows


class PSNetwork(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.conf = conf
        self.render_model = conf.get_string('train.render_model', default='sgbasis')
        if self.render_model not in ('sgbasis', 'microfacet'):
            raise ValueError('unknown train.render_model %r' % self.render_model)
        self.n_freqs = conf.get_int('brdf.net.n_freqs_xyz')
        dim_emb = 3 + 6 * self.n_freqs
        assert dim_emb <= PE_STRIDE
        W, depth = conf.get_int('brdf.net.mlp_width'), conf.get_int('brdf.net.mlp_depth')
        skip = conf.get_int('brdf.net.mlp_skip_at')
        if self.render_model == 'microfacet':  # renderer.py:62-63,74-75
            self.f0 = conf.get_float('brdf.fresnel_f0', default=0.05)
            self.albedo_net = MLP(dim_emb, 3, W, depth, skip_at=[skip], final='sigmoid')
            self.rough_net = MLP(dim_emb, 1, W, depth, skip_at=[skip], final='sigmoid')
        else:
            nbasis = conf.get_int('train.nbasis', default=9)
            self.specular_rgb = conf.get_bool('train.specular_rgb', default=False)
            self.sgbasis = SGBasis(nbasis=nbasis, specular_rgb=self.specular_rgb)
            self.albedo_net = MLP(dim_emb, 3, W, depth, skip_at=[skip], final='sigmoid')
            if self.specular_rgb:
                nbasis *= 3
            self.rough_net = MLP(dim_emb, nbasis, conf.get_int('brdf.sgnet.mlp_width', 128),
                                 conf.get_int('brdf.sgnet.mlp_depth', 4),
                                 skip_at=[conf.get_int('brdf.sgnet.mlp_skip_at', 2)])
            self.nbasis = nbasis
        self.light_int = conf.get_float('brdf.light_intensity', default=4.0)
        self.shape_pregen = conf.get_bool('train.shape_pregen', default=False)
        if not self.shape_pregen:
            raise NotImplementedError('train.shape_pregen=False (the reference leaves surface_points undefined there)')
        self.xyz_jitter_std = conf.get_float('brdf.net.xyz_jitter_std', default=0)
        self.normal_mlp = conf.get_bool('train.normal_mlp', default=False)
        if self.normal_mlp:
            self.n_freqs_n = conf.get_int('normal.net.n_freqs_xyz')
            self.normal_net = MLP(3 + 6 * self.n_freqs_n, 3, conf.get_int('normal.net.mlp_width'),
                                  conf.get_int('normal.net.mlp_depth'),
                                  skip_at=[conf.get_int('normal.net.mlp_skip_at')])
            self.normal_joint = conf.get_bool('train.normal_joint', default=False)
            self.normal_jitter_std = conf.get_float('normal.net.xyz_jitter_std', default=0)
            if not self.normal_joint:
                self.normal_net = self.normal_net.eval().requires_grad_(False)
                self.normal_jitter_std = 0
        # 'fp32' (default, exact) or 'bf16': gradient-free visibility_net evaluations (evaluation / relighting,
        # stage2/eval.py:199-218) on the bf16 MFMA engine.  Never used while gradients are enabled.
        self.inference_precision = 'fp32'
        self._eval_cache = None  # dict while an evaluation loop over light batches is running (see _memo)
        self._eval_outputs = None  # set of output keys while an evaluation loop wants only those (see the end of forward)
        # opt-in, NOT the reference's arithmetic (default off; every parity test and the headline benchmark run without
        # it): during training, the L shading-light visibility rows -- which only enter the loss detached,
        # renderer.py:197 -- are evaluated on the bf16 engine (max |d| ~ 2e-3 on the visibility value).
        self.train_vis_bf16 = conf.get_bool('train.vis_bf16', default=False)
        # opt-in EXPERIMENT (default off; never the headline): the same rows on the split-bf16 engine (csrc/mlp_infer_x3.hip),
        # i.e. fp32-class arithmetic -- every operand as three bf16 planes, six partial products per multiply -- on the bf16
        # matrix pipe.  inference_precision = 'bf16x6' selects it for gradient-free evaluations.
        self.train_vis_bf16x6 = conf.get_bool('train.vis_bf16x6', default=False)
        # ... or through the exact engine's own kernel on split-bf16 weight stages (two bf16 pieces per operand, three partial
        # products, ~1e-5 relative: PsnMlpDesc.w_format = PSN_W_BF16X2); inference_precision = 'bf16x3' for evaluations
        self.train_vis_bf16x3 = conf.get_bool('train.vis_bf16x3', default=False)
        # The BRDF / normal networks of a training forward (and, through autograd, their backward) run on a side stream
        # beside the visibility launch (20 of the 27 ms of a step; it is issued first and needs none of their results):
        # Ns-row launches of 25 - 70 us each, latency-bound on their own, fill the gaps of the big launch instead of
        # following it.  Off: everything on the caller's stream.
        self.overlap_small_nets = conf.get_bool('train.overlap_small_nets', default=True)
        self._side = {}
        self._cols_cache = {}
        self.visibility = conf.get_bool('train.visibility', default=False)
        self.light_vis_detach = conf.get_bool('train.light_vis_detach', default=False)
        if self.visibility:
            self.visibility_net = MLP(dim_emb * 2, 1, conf.get_int('visibility.net.mlp_width'),
                                      conf.get_int('visibility.net.mlp_depth'),
                                      skip_at=[conf.get_int('visibility.net.mlp_skip_at')])

    # -- helpers ---------------------------------------------------------------------------------
    def invalidate_packs(self, trainable_only=False):
        """Drop every cached weight pack of the model (call after editing parameters through ``.data``: such edits
        change neither the version counters nor the storage addresses the caches are keyed on).  trainable_only: keep the
        packs of networks without a trainable parameter (what TrainStep calls after every optimiser step: a fused
        optimiser implementation does not bump version counters either)."""
        mlps = self.__dict__.get('_mlp_list')
        if mlps is None:  # (the module tree is fixed after construction: walked once, not once per optimiser step)
            mlps = self.__dict__['_mlp_list'] = [m for m in self.modules() if isinstance(m, MLP)]
        live = [m for m in mlps if not trainable_only or any(q.requires_grad for q in m.parameters())]
        if live:
            self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1  # (the model-level epoch keys the visibility packs)
        for m in live:
            m.invalidate_packs()

    def _apply(self, fn, *a, **k):
        self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1
        return super()._load_from_state_dict(*a, **k)

    def _pe(self, x, n_freqs):
        return ops.positional_encoding(x, n_freqs, PE_STRIDE)

    def _cols(self, n_freqs, device, pair=False):
        key = (n_freqs, str(device), pair)  # constant index lists: built once (4 launch-bound kernels per call otherwise)
        c = self._cols_cache.get(key)
        if c is None:
            d = 3 + 6 * n_freqs
            c = torch.arange(d, device=device)
            c = self._cols_cache[key] = torch.cat([c, PE_STRIDE + c]) if pair else c
            c._psn_contiguous_pair = bool(pair)  # (ops.VisibilityPair: the two halves are the leading columns of their tables)
            if c.is_cuda:
                torch.cuda.current_stream(c.device).synchronize()  # read on both streams from now on: written before either does
        return c

    def _visibility_rows(self, pe_x, light_dirs, fused_ok):
        """visibility_net on the light-major rows [pe_x[n] | PE(light[l])], l-major (renderer.py:191-200)."""
        l_in = light_dirs.detach() if self.light_vis_detach else light_dirs
        pe_l = self._pe(l_in, self.n_freqs)
        net = self.visibility_net
        Ws, bs = net.weights()
        cols = self._cols(self.n_freqs, pe_x.device, pair=True)
        if (fused_ok and net.width == 256 and self.inference_precision == 'bf16x3' and not torch.is_grad_enabled() and pe_x.is_cuda):
            return self._visibility_rows_b3(pe_x, pe_l)
        if (fused_ok and net.width == 256 and self.inference_precision in ('bf16', 'bf16x6') and not torch.is_grad_enabled()
                and len(Ws) <= 12 and pe_x.is_cuda):
            # opt-in bf16 MFMA engines (evaluation / relighting): plain bf16 (csrc/mlp_infer_bf16.hip) or the split form with
            # fp32-class accuracy (csrc/mlp_infer_x3.hip)
            if self.inference_precision == 'bf16x6':
                return self._visibility_rows_x3(pe_x, pe_l)
            return self._visibility_rows_bf16(pe_x, pe_l)
        if fused_ok and net.width == 256:
            params = []
            for W, b in zip(Ws, bs):
                params += [W, b]
            return ops.FusedPairMLP.apply(pe_x, pe_l, cols, net._skip_index(), *params)
        if fused_ok:
            ops.fallback('stage2.visibility rows -> expanded [L Ns, 128] input', pe_x, 'visibility_net width %d' % net.width)
        ns, nl = pe_x.shape[0], pe_l.shape[0]
        x = torch.cat([pe_x.tile(nl, 1), pe_l.repeat_interleave(ns, dim=0)], dim=1)
        return net(x, cols)

    def _visibility_pair_args(self, pe_x, light_dir, light_vis_train):
        pe_l = self._pe(torch.cat([light_dir.detach(), light_vis_train.detach()], dim=0), self.n_freqs)
        net = self.visibility_net
        Ws, bs = net.weights()
        params = []
        for W, b in zip(Ws, bs):
            params += [W, b]
        cols = self._cols(self.n_freqs, pe_x.device, pair=True)
        return pe_x.detach(), pe_l, light_dir.shape[0], cols, net._skip_index(), params

    def _visibility_prepack(self):
        """Packed visibility-net weights (input block through init tables), rebuilt only when the parameters changed."""
        net = self.visibility_net
        if net.width != 256 or not net.linears[0].weight.is_cuda:
            return None
        key = params_key(net.parameters(), getattr(self, '_pack_epoch', 0))
        if getattr(self, '_vis_pack_key', None) != key:
            Ws, bs = net.weights()
            half = 3 + 6 * self.n_freqs
            with torch.no_grad():
                # (the previous pack's init-table buffers are rewritten in place: its launches are behind us on this stream)
                self._vis_pack = fused.pack_relu_mlp(list(Ws), list(bs), half, half, net._skip_index(),
                                                     reuse=getattr(self, '_vis_pack', None))
            self._vis_pack_key = key
        return self._vis_pack

    @torch.no_grad()
    def _visibility_rows_bf16(self, pe_x, pe_l):
        """Gradient-free visibility_net rows (light-major) on the bf16 MFMA engine."""
        # grouped form: the light's half of the input block enters as a per-light bias (fp32 product, once per light)
        return self._visibility_prepack_bf16()(pe_x.to(torch.bfloat16), pe_l.contiguous())

    @torch.no_grad()
    def _visibility_rows_x3(self, pe_x, pe_l):
        """Gradient-free visibility_net rows (light-major) on the split-bf16 engine (fp32-class accuracy; experiment)."""
        net = self.visibility_net
        key = params_key(net.parameters(), getattr(self, '_pack_epoch', 0))
        if getattr(self, '_vis_packx3_key', None) != key:
            Ws, bs = net.weights()
            half = 3 + 6 * self.n_freqs
            self._vis_packx3 = fused.pack_relu_mlp_x3_grouped(list(Ws), list(bs), half, half, net._skip_index(),
                                                              hip.OUT_SIGMOID if net.final == 'sigmoid' else hip.OUT_NONE)
            self._vis_packx3_key = key
        return self._vis_packx3(pe_x.contiguous(), pe_l.contiguous())

    @torch.no_grad()
    def _visibility_rows_b3(self, pe_x, pe_l):
        """Gradient-free visibility_net rows (light-major) through the exact engine's kernel on split-bf16 weight stages (experiment)."""
        net = self.visibility_net
        key = params_key(net.parameters(), getattr(self, '_pack_epoch', 0))
        if getattr(self, '_vis_packb3_key', None) != key:
            Ws, bs = net.weights()
            half = 3 + 6 * self.n_freqs
            self._vis_packb3 = fused.pack_relu_mlp(list(Ws), list(bs), half, half, net._skip_index(),
                                                   out_act=hip.OUT_SIGMOID if net.final == 'sigmoid' else hip.OUT_NONE,
                                                   reuse=getattr(self, '_vis_packb3', None), x3=True)
            self._vis_packb3_key = key
        ns, nl = pe_x.shape[0], pe_l.shape[0]
        return self._vis_packb3(pe_x.contiguous(), nl * ns, a_div=1, a_mod=ns, tab_b=pe_l.contiguous(), b_div=ns, b_mod=nl)

    def _memo(self, tag, input, fn):
        """Light-independent intermediate of a gradient-free evaluation, computed once per (pixel set, weights) while a
        caller holds ``self._eval_cache`` open (relight.render_envmap evaluates the same pixels under 8 light batches:
        positional encodings and the BRDF / normal nets do not depend on the lights).  Otherwise just fn()."""
        cache = self._eval_cache
        if cache is None or torch.is_grad_enabled():
            return fn()
        key = (tag,) + tuple((input[k].data_ptr(), input[k]._version) for k in ('points', 'surface_mask', 'uv')) \
            + params_key(self.parameters(), getattr(self, '_pack_epoch', 0))
        if key not in cache:
            cache[key] = fn()
        return cache[key]

    def _visibility_prepack_bf16(self):
        """visibility-net weights in the fragment order of the bf16 engine, rebuilt only when the parameters changed."""
        net = self.visibility_net
        key = params_key(net.parameters(), getattr(self, '_pack_epoch', 0))
        if getattr(self, '_vis_pack16_key', None) != key:
            Ws, bs = net.weights()
            half = 3 + 6 * self.n_freqs
            with torch.no_grad():
                self._vis_pack16 = fused.pack_relu_mlp_bf16_grouped(list(Ws), list(bs), half, half, net._skip_index(),
                                                                    hip.OUT_SIGMOID if net.final == 'sigmoid' else hip.OUT_NONE)
            self._vis_pack16_key = key
        return self._vis_pack16

    def _visibility_pair_launch(self, pe_x, light_dir, light_vis_train, live_count=None):
        """Issue the fused launch now, attach the autograd node later (ops.VisibilityPair.launch).  live_count: the device-side
        number of real rows of a padded surface list ('surface_count' of the batch)."""
        a = self._visibility_pair_args(pe_x, light_dir, light_vis_train)
        need = any(p.requires_grad for p in a[5])
        return a, ops.VisibilityPair.launch(a[0], a[1], a[2], a[3], a[4], [p.detach() for p in a[5]], need,
                                            packed=self._visibility_prepack(), live_count=live_count)

    def _visibility_pair(self, pe_x, light_dir, light_vis_train, launched=None):
        """(vis [L*Ns,1], vis_train [V*Ns,1]) from one fused launch (ops.VisibilityPair)."""
        if launched is not None:
            a, pre = launched
        else:
            a, pre = self._visibility_pair_args(pe_x, light_dir, light_vis_train), None
        return ops.VisibilityPair.apply(a[0], a[1], a[2], a[3], a[4], pre, *a[5])

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, input, albedo_new=None, basis_new=None, noise=None):
        """Same inputs / outputs as the reference forward (renderer.py:110-266).  The reference gathers and
        scatters with boolean masks (``x[surface_mask]``, ``dense[mask.expand(L,-1)] = v``), each of which costs a
        nonzero() + host sync on a GPU; here the surface index list is computed once and every gather / scatter
        is an index_select / index_copy with it (identical element order: ascending pixel index, light-major)."""
        noise = noise or {}
        uv, pose, intr = input['uv'], input['pose'], input['intrinsics']
        object_mask = input['object_mask']
        device = uv.device
        surface_mask, points, normals = input['surface_mask'], input['points'], input['normal']
        # The index list of the surface pixels: nonzero() is the one data-dependent host synchronisation of the forward.
        # A batch may bring the list along ('surface_idx', int64, ascending = surface_mask[0].nonzero()): the mask is an
        # INPUT (stage-1 hand-off data), so the data pipeline -- handoff.ViewSampler on the host, where the mask lives
        # anyway -- can build it while the previous step is still running, and the host then queues a whole step
        # without ever waiting for the GPU.
        idx = input.get('surface_idx')
        if idx is None:
            idx = surface_mask[0].nonzero(as_tuple=True)[0]
        ns = idx.shape[0]

        def gather(t):  # [1,N,C] -> [Ns,C]
            return t[0].index_select(0, idx)

        # Dense outputs are described first and written at the end, ALL of them by one launch (ops.ScatterRows): a
        # _Dense is "B x N x C filled with `fill`", optionally carrying surface rows [B*Ns, C] (light-major).
        n_pix = points.shape[1]

        def scatter(dense, rows):
            return _Dense(dense.fill, dense.B, dense.C, rows)

        def ones3(b=1):
            return _Dense(1.0, b, 3)

        surf = gather(points).contiguous()
        out_n = {}
        normal_s = None
        # The fused visibility launch is 60 % of the step and depends only on the surface points and the light
        # directions: issue it FIRST, so the GPU is busy while the host queues the small BRDF / normal-net launches
        # (they then run back to back behind it instead of each waiting for its own launch latency).  Its autograd
        # node is attached at the reference's position in the graph, further down.
        pe_x = None
        vis_pair = None
        vis_bf16 = None
        side = None
        if ns > 0 and self.visibility:
            lv0 = input.get('light_vis_train')
            ld0 = input['light_direction']
            if (lv0 is not None and self.conf.get_bool('train.vis_rgb_detach', default=False)
                    and self.visibility_net.width == 256 and torch.is_grad_enabled()
                    and (self.light_vis_detach or not (ld0.requires_grad or lv0.requires_grad))):
                pe_x = self._pe(surf, self.n_freqs)
                if self.overlap_small_nets and surf.is_cuda:
                    # fork BEFORE the big launch is queued: the side stream waits for everything issued so far (inputs,
                    # encodings, the optimiser step of the previous iteration), not for the visibility kernel
                    side = self._side.get(device)
                    if side is None:
                        side = self._side[device] = torch.cuda.Stream(device=device)  # (a high-priority stream: +-0.05 ms/step at 32768 px; round 5 at 4096 px: 3.76 -> 3.78 ms, 32768 px eager 25.1 -> 27.1 ms)
                    side.wait_stream(torch.cuda.current_stream(device))
                if self.train_vis_bf16 or self.train_vis_bf16x6 or self.train_vis_bf16x3:
                    # opt-in (train.vis_bf16 / train.vis_bf16x6): the L shading rows enter the loss detached (renderer.py:197),
                    # so they can run on a bf16 engine; the V supervised rows stay on the exact fp32 path with their dumps
                    pe_l0 = self._pe(ld0.detach(), self.n_freqs)
                    vis_bf16 = (self._visibility_rows_b3(pe_x, pe_l0) if self.train_vis_bf16x3 else
                                self._visibility_rows_x3(pe_x, pe_l0) if self.train_vis_bf16x6 else self._visibility_rows_bf16(pe_x, pe_l0))
                    vis_pair = self._visibility_pair_launch(pe_x, ld0[:0], lv0)
                else:
                    # 'surface_count' (float32 [1] on the device, beside a 'surface_idx' padded to a fixed capacity -- hip.surface_index,
                    # GraphedTrainStep(pad_to_pixels=True)): the gradient-free shading rows of the padding are not evaluated
                    vis_pair = self._visibility_pair_launch(pe_x, ld0, lv0, live_count=input.get('surface_count'))
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        if self.normal_mlp:  # renderer.py:127-143
            normal_pred = ones3()
            if ns > 0:
              with on_side():
                cols_n = self._cols(self.n_freqs_n, device)
                # (the encoding of the surface points is shared with the visibility / BRDF networks when the octave counts agree)
                pe_n = (lambda: pe_x) if (pe_x is not None and self.n_freqs_n == self.n_freqs) else (lambda: self._pe(surf, self.n_freqs_n))
                if self.normal_jitter_std > 0 and torch.is_grad_enabled():
                    # base and jittered evaluation (renderer.py:127-143) in the same launches, as the BRDF networks below: rows
                    # [0, Ns) = PE(x), rows [Ns, 2 Ns) = PE(x + noise).  Row-wise identical results; one forward launch, one
                    # backward chain and one weight-gradient launch instead of two each (they are latency-bound at Ns rows)
                    nz = noise.get('normal')
                    if nz is None:
                        nz = torch.empty_like(surf).normal_(0.0, self.normal_jitter_std)  # = torch.normal(0, std) without its host-side std >= 0 check (one launch)
                    pe_both_n = torch.cat([pe_n(), self._pe(surf + nz, self.n_freqs_n)], dim=0)
                    normal_s, nj = ops.SplitRows.apply(ops.normalize_rows(self.normal_net(pe_both_n, cols_n)), ns)
                    normal_pred = scatter(normal_pred, normal_s)
                    out_n['normal_jitter'] = scatter(ones3(), nj)
                else:
                    normal_s = self._memo('normal', input, lambda: ops.normalize_rows(self.normal_net(pe_n(), cols_n)))
                    normal_pred = scatter(normal_pred, normal_s)
                    if self.normal_jitter_std > 0:
                        nz = noise.get('normal')
                        if nz is None:
                            nz = torch.empty_like(surf).normal_(0.0, self.normal_jitter_std)
                        nj = self._memo('normal_jitter', input, lambda: ops.normalize_rows(
                            self.normal_net(self._pe(surf + nz, self.n_freqs_n), cols_n)))
                        out_n['normal_jitter'] = scatter(ones3(), nj)
            out_n['normal_pred'] = normal_pred

        sg = self.render_model == 'sgbasis'
        lnum = input['light_direction'].shape[0]
        rgb_values = ones3(lnum)
        albedo_values = ones3()
        rough_values = ones3(lnum) if sg else ones3()
        weight_values = _Dense(0.0, 1, self.nbasis) if sg else None
        vis_values = ones3(lnum)
        jitter = None
        vis_t_pre = None
        if ns > 0:
            normal = gather(normals) if not self.normal_mlp else normal_s
            # camera rays (~15 launch-bound elementwise kernels) are only needed by the shading: queued beside the
            # visibility launch (side stream), not in front of it
            with on_side():
                if (uv.is_cuda and uv.shape[0] == 1 and uv.dtype == torch.float32 and tuple(pose.shape[1:]) == (4, 4)
                        and tuple(intr.shape[1:]) == (4, 4)):  # (psn_camera_rays reads the 4x4 layouts; a 3x3 K takes the torch form)
                    # rays of the surface pixels only, normalised and negated, in one launch (psn_camera_rays)
                    pts2c = hip.camera_rays(uv.contiguous(), pose.contiguous().float(), intr.contiguous().float(), idx, scale=-1.0)
                else:
                    ray_dirs, _ = camera_rays(uv, pose, intr)
                    pts2c = -gather(ray_dirs)
            light_dir = input['light_direction']
            cols = self._cols(self.n_freqs, device)
            if pe_x is None:
                pe_x = self._memo('pe_x', input, lambda: self._pe(surf, self.n_freqs))
            # The jittered re-evaluation of the BRDF nets (renderer.py:211-231) rides in the same launches as the base
            # evaluation: rows [0, Ns) = PE(x), rows [Ns, 2 Ns) = PE(x + noise).  Row-wise identical results, half the
            # (latency-bound, Ns-row) GEMM launches in forward and backward.
            pe_j = None
            with on_side():
              if self.xyz_jitter_std > 0:
                nz = noise.get('xyz')
                if nz is None:
                    nz = torch.empty_like(surf).normal_(0.0, self.xyz_jitter_std)  # = torch.normal(0, std) without its host-side std >= 0 check (one launch)
                def brdf_both():
                    pe_both = torch.cat([pe_x, self._pe(surf + nz, self.n_freqs)], dim=0)
                    return self.albedo_net(pe_both, cols), self.rough_net(pe_both, cols)
                albedo_both, rough_both = self._memo('brdf_both', input, brdf_both)
                albedo, albedo_j = ops.SplitRows.apply(albedo_both, ns)
                if sg:
                    # SG weights = relu(network output) for both halves (renderer.py:180, 224): ONE clamp on the 2 Ns rows in
                    # front of the split instead of one per half (and one threshold_backward instead of two)
                    rough_both = F.relu(rough_both)
                rough, rough_j = ops.SplitRows.apply(rough_both, ns)
              else:
                albedo, rough = self._memo('brdf', input, lambda: (self.albedo_net(pe_x, cols), self.rough_net(pe_x, cols)))
              if albedo_new is not None:
                albedo = torch.from_numpy(albedo_new).to(device)[None].expand_as(albedo)
              if sg:
                weights = rough if self.xyz_jitter_std > 0 else F.relu(rough)  # (already clamped in front of the split)
                if basis_new is not None:  # material editing (eval.py:233-312)
                    wn = torch.zeros_like(weights)
                    if self.specular_rgb:
                        wn.view(-1, 3, self.nbasis // 3)[:, :, basis_new] = 2 ** basis_new / 100
                    else:
                        wn.view(-1, 1, self.nbasis)[:, :, basis_new] = 2 ** basis_new / 100
                    weights = wn.reshape(-1, self.nbasis)
                weight_values = scatter(weight_values, weights)
            if side is not None:
                # join: everything below (shading, dense outputs, losses) reads the small networks' outputs on the caller's
                # stream.  Their memory came from the side stream's pool: tell the allocator about the second user.
                main = torch.cuda.current_stream(device)
                main.wait_stream(side)
                for t_ in (pts2c, normal_s, albedo_both if self.xyz_jitter_std > 0 else albedo, rough_both if self.xyz_jitter_std > 0 else rough,
                           weights if sg else None, out_n.get('normal_jitter').rows if 'normal_jitter' in out_n else None):
                    if torch.is_tensor(t_):
                        t_.record_stream(main)
            light_int = input.get('light_intensity', self.light_int)
            vis = None
            vis_for_rgb = None
            vis_t_pre = None
            if self.visibility:
                detach = self.conf.get_bool('train.vis_rgb_detach', default=False)
                lv = input.get('light_vis_train')
                pair_ok = (lv is not None and detach and self.visibility_net.width == 256 and torch.is_grad_enabled()
                           and (self.light_vis_detach or not (light_dir.requires_grad or lv.requires_grad)))
                if pair_ok:
                    # shading rows and supervision rows in ONE fused launch (the latter dump their activations)
                    # (round 6, measured and dropped: attaching this node BEHIND the shading node -- so that autograd queues the V-row
                    #  chain and its 256 x 256 weight gradients first -- costs 0.23 ms at 4096 px (3.74 -> 3.97 ms replayed) and 0.45 ms
                    #  at 32768 px: the weight-gradient launch holds every CU while the small networks' backward, the longer dependent
                    #  chain, waits; profiles/r06d_ab_strong4096.jsonl)
                    if vis_bf16 is not None:
                        _none, vis_t_pre = self._visibility_pair(pe_x, light_dir[:0], lv, launched=vis_pair)
                        vis = vis_bf16
                    else:
                        vis, vis_t_pre = self._visibility_pair(pe_x, light_dir, lv, launched=vis_pair)
                else:
                    # gradient-free unless a caller backpropagates into output['visibility']
                    vis = self._visibility_rows(pe_x, light_dir, fused_ok=True)  # [L*Ns, 1], light-major
                vis_for_rgb = vis.detach() if detach else vis
            if sg:
                rgb, spec = ops.sg_shade(light_dir, pts2c, normal, albedo, weights, self.sgbasis.lobe, light_int,
                                         vis_for_rgb, self.specular_rgb)
            else:
                rgb = ops.mf_shade(light_dir, pts2c, normal, albedo, rough, light_int, vis_for_rgb, self.f0)
            rgb_values = scatter(rgb_values, rgb)
            if vis is not None:
                vis_values = scatter(vis_values, vis.expand(rgb.shape))
            albedo_values = scatter(albedo_values, albedo)
            if sg:
                rough_values = scatter(rough_values, spec.expand(-1, 3))
            else:
                rough_values = scatter(rough_values, rough.expand(-1, 3))
            if self.xyz_jitter_std > 0:  # renderer.py:211-231
                aj = scatter(ones3(), albedo_j)
                if sg:
                    rj = scatter(_Dense(1.0, 1, self.nbasis), rough_j)  # (= relu of the jittered evaluation, clamped above)
                    r_ori = weight_values
                else:
                    rj = scatter(ones3(), rough_j.expand(-1, 3))
                    r_ori = rough_values
                jitter = {'albedo_values': albedo_values, 'albedo_jitter': aj,
                          'rough_values': r_ori, 'rough_jitter': rj}

        out = {
            'points': points, 'object_mask': object_mask, 'network_object_mask': surface_mask,
            'sg_rgb_values': rgb_values, 'normal_values': normals,
            'sg_diffuse_albedo_values': albedo_values, 'sg_specular_rgb_values': rough_values,
        }
        if jitter is not None:
            out.update(jitter)
        if self.normal_mlp:
            out.update(out_n)
        if self.visibility:
            out['visibility'] = vis_values
            if 'vis_train_gt' in input or 'light_vis_train' in input:  # renderer.py:251-262
                lv = input['light_vis_train']
                vnum = lv.shape[0]
                vt = ones3(vnum)
                if ns > 0:
                    if vis_t_pre is not None:
                        vis_t = vis_t_pre
                    else:
                        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.visibility_net.parameters())
                        vis_t = self._visibility_rows(pe_x, lv, fused_ok=not train)
                    vt = scatter(vt, vis_t.expand(-1, 3))
                out['vis_train'] = vt
        if sg:
            out['sg_weight'] = weight_values
        want = self._eval_outputs
        if want is not None:
            # an evaluation loop (relight.render_envmap) that consumes a few outputs only: the others are never written, and
            # 'sg_rgb_light_sum' = (sum over this batch's lights of the surface rows [Ns, 3], idx [Ns], the constant every
            # other pixel sums to) replaces the dense [L, N, 3] tensor + its reduction (eval.py:218 sums the lights anyway)
            if 'sg_rgb_light_sum' in want:
                v = out['sg_rgb_values']
                rows_sum = v.rows.reshape(v.B, -1, v.C).sum(0) if v.rows is not None else None
                out['sg_rgb_light_sum'] = (rows_sum, idx if v.rows is not None else None, v.fill * v.B)
            out = {k: v for k, v in out.items() if k in want}
        # write every dense output: one launch for those that carry surface rows, constant fills for the others
        lazy = {}  # id -> _Dense (one object may sit under two keys: 'rough_values' of the jitter dict is 'sg_weight')
        for v in out.values():
            if isinstance(v, _Dense):
                lazy[id(v)] = v
        with_rows = [v for v in lazy.values() if v.rows is not None]
        done = {}
        if with_rows:
            if idx.is_cuda and idx.dtype == torch.int64:
                inv = hip.inverse_index(idx.contiguous(), n_pix, count=input.get('surface_count'))  # pixel -> surface row or -1, one launch (idx is ascending)
            else:
                inv = torch.full((n_pix,), -1, dtype=torch.int32, device=device)
                inv[idx] = torch.arange(ns, dtype=torch.int32, device=device)
            specs = tuple((v.B, v.C, v.fill) for v in with_rows)
            dense = ops.ScatterRows.apply(idx, inv, specs, *[v.rows for v in with_rows])
            for v, d in zip(with_rows, dense):
                done[id(v)] = d
        for v in lazy.values():
            if v.rows is None:
                done[id(v)] = torch.full((v.B, n_pix, v.C), v.fill, device=device)
        for k in list(out.keys()):
            if isinstance(out[k], _Dense):
                out[k] = done[id(out[k])]
        return out
