"""torch.autograd.Function wrappers over the C ABI (psnerf_amd.hip).

PyTorch is used here only as the owner of device memory and the autograd tape;
every FLOP on the hot path runs in libpsnerf_hip.so.  Each Function implements
its own backward with the same kernels (no autograd-of-autograd).
"""
import collections

import torch

from . import fused, hip


# How often each autograd node's FORWARD ran, by class name (host-side, no device work): tests assert on it that a step took the
# fused engines ('VisibilityPair', 'FusedReluNet', 'GeoFieldFused', 'AppNetFused', ...) and not the layer-wise GEMM
# formulations kept for other network widths ('ReluMLP', 'GeoField') -- a shape predicate that quietly fails on the GPU would
# otherwise fall back without anybody noticing.
HITS = collections.Counter()


def _hit(name):
    HITS[name] += 1


# Engine selection made loud.  Every place where a DEVICE tensor is about to leave the register-resident /
# fused engines for a layer-wise GEMM sequence or a torch formulation -- an unsupported width, a shape or dtype predicate that
# failed -- calls ``fallback(name, tensor, why)``: it is counted in FALLBACKS and, when STRICT is on (bench.py, tools/run_e2e.py,
# the operating-point tests: ``with ops.strict():``; or PSN_STRICT=1), raises instead of silently taking the slow path.  CPU
# tensors (module construction, checkpoint loading, the CPU test-suite) and switches a caller set deliberately
# (MLP.FUSED = False, Renderer.FUSED_SWEEP = False, ... -- the A/B and cross-check paths of the tests) are not fallbacks.
import os as _os0
STRICT = _os0.environ.get('PSN_STRICT', '0') == '1'
FALLBACKS = collections.Counter()


def fallback(name, tensor=None, why=''):
    if tensor is not None and not getattr(tensor, 'is_cuda', False):
        return
    FALLBACKS[name] += 1
    if STRICT:
        raise RuntimeError('psnerf_amd.ops.STRICT: %s left the fused engines on a device tensor%s' % (name, (' (' + why + ')') if why else ''))


class strict(object):
    """``with ops.strict():`` -- a device tensor that falls off the fused engines raises (ops.fallback) instead of falling back."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global STRICT
        self.saved, STRICT = STRICT, self.on
        return self

    def __exit__(self, *exc):
        global STRICT
        STRICT = self.saved
        return False


def reset_hits():
    HITS.clear()
    FALLBACKS.clear()


def _rowsum_small(t):
    """Column sums [C] of a contiguous [Q, C] tensor with C <= 4 (bias gradients of 1- / 3-output heads).  torch's
    dim-0 reduction of such a shape is pathological (250 us for [524288, 3]); viewed as [Q C / 64 C, 64 C] it is a
    16-byte-load column sum (psn_colsum) followed by a 64 x C fold."""
    Q, C = t.shape
    W = 64 * C
    if t.is_cuda and t.is_contiguous() and C <= 4 and Q >= 4096 and (Q * C) % W == 0:
        return hip.colsum(t.view(-1, W)).view(64, C).sum(0)
    return t.sum(0)


def _split_k_for(rows, out_rows=256, out_cols=256):
    """Split-K factor for a weight-gradient GEMM [out_rows, rows] x [rows, out_cols]: enough K slices that
    (output tiles x slices) fills the 256 CUs a few times over, but at least 256 rows per slice."""
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    want = max(1, 1024 // tiles)
    return int(max(1, min(want, rows // 256)))


# --------------------------------------------------------------------------- positional encoding
class PositionalEncoding(torch.autograd.Function):
    """[x, sin(2^k x), cos(2^k x)]_k zero-padded to ``out_stride`` columns.
    stage1/model/network.py:141-150 and stage2/model/embedder.py:6-54."""

    @staticmethod
    def forward(ctx, x, n_freqs, out_stride, scale=1.0):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.n_freqs, ctx.scale = n_freqs, scale
        return hip.pe_encode(x, n_freqs, out_stride, scale)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        return hip.pe_encode_bwd(x, d_out.contiguous(), ctx.n_freqs, ctx.scale), None, None, None


def positional_encoding(x, n_freqs, out_stride=None, scale=1.0):
    width = 3 + 6 * n_freqs
    return PositionalEncoding.apply(x, n_freqs, width if out_stride is None else out_stride, scale)


# --------------------------------------------------------------------------- alpha composite
class AlphaComposite(torch.autograd.Function):
    """stage1/model/rendering.py:196-197,214-216: returns (rgb [N,3] incl. white background, acc [N]).  The per-sample
    weights are an intermediate of the reference (not part of its output dictionary) and the backward kernel rebuilds
    them from alpha, so they are not written: 16 S + 16 B per ray instead of 20 S + 16."""

    @staticmethod
    def forward(ctx, alpha, rgb, white_bg):
        _hit('AlphaComposite')
        alpha, rgb = alpha.contiguous(), rgb.contiguous()
        _w, out, acc = hip.composite_fwd(alpha, rgb, white_bg, need_weights=False)
        ctx.save_for_backward(alpha, rgb)
        ctx.white_bg = white_bg
        return out, acc

    @staticmethod
    def backward(ctx, d_out, d_acc):
        alpha, rgb = ctx.saved_tensors
        d_out = torch.zeros(alpha.shape[0], 3, device=alpha.device) if d_out is None else d_out.contiguous()
        d_acc = None if d_acc is None else d_acc.contiguous()
        da, dc = hip.composite_bwd(alpha, rgb, d_out, d_acc, ctx.white_bg)
        return da, dc, None


def alpha_composite(alpha, rgb, white_bg):
    return AlphaComposite.apply(alpha, rgb, white_bg)


# --------------------------------------------------------------------------- ReLU MLP (stage2 nets)
def _padded_weight(W, cols, kp):
    """Scatter the columns of W [o, k] to positions ``cols`` of a zero [o, kp] matrix."""
    if cols is None:
        return W.contiguous()
    Wp = torch.zeros(W.shape[0], kp, device=W.device, dtype=W.dtype)
    Wp[:, cols] = W
    return Wp


class ReluMLP(torch.autograd.Function):
    """stage2 Network / Normal_Network (stage2/model/renderer.py:17-49) for training: Linear+ReLU stack,
    the input is concatenated after layer ``skip_at``; final layer linear or sigmoid.

    x: [Q, kp] padded input features; ``in_cols`` maps the din real input columns into the kp padded
    ones (PE tables are padded to multiples of 32/4 floats so every operand is 16-byte aligned).
    params: W0, b0, W1, b1, ...  (torch Linear layout [out, in])."""

    @staticmethod
    def forward(ctx, x, in_cols, skip_at, final_sigmoid, *params):
        _hit('ReluMLP')
        n = len(params) // 2
        Ws, bs = params[0::2], params[1::2]
        Q, kp = x.shape
        width = Ws[0].shape[0]
        need_grad = any(ctx.needs_input_grad)
        saved_in = []
        Wps = []
        h = x
        for li in range(n):
            last = li == n - 1
            if li == 0:
                Wp = _padded_weight(Ws[0], in_cols, kp)
            elif li - 1 == skip_at:
                cols = torch.cat([torch.arange(width, device=x.device), width + in_cols])
                Wp = _padded_weight(Ws[li], cols, width + kp)
            else:
                Wp = Ws[li].contiguous()
            Wps.append(Wp)
            saved_in.append(h)
            if last:
                out = hip.gemm(h, Wp, trans_b=True, bias=bs[li].contiguous(),
                               epi=hip.EPI_BIAS_SIGMOID if final_sigmoid else hip.EPI_BIAS)
            elif li == skip_at:
                buf = torch.empty(Q, width + kp, device=x.device, dtype=torch.float32)
                hip.gemm(h, Wp, trans_b=True, bias=bs[li].contiguous(), epi=hip.EPI_BIAS_RELU, out=buf[:, :width])
                buf[:, width:] = x
                h = buf
            else:
                h = hip.gemm(h, Wp, trans_b=True, bias=bs[li].contiguous(), epi=hip.EPI_BIAS_RELU)
        if need_grad:
            ctx.save_for_backward(out, in_cols, *saved_in, *Wps)
            ctx.n, ctx.skip_at, ctx.final_sigmoid, ctx.width, ctx.kp = n, skip_at, final_sigmoid, width, kp
            ctx.x_needs = ctx.needs_input_grad[0]
        return out

    @staticmethod
    def backward(ctx, d_out):
        n, width, kp = ctx.n, ctx.width, ctx.kp
        saved = ctx.saved_tensors
        out, in_cols = saved[0], saved[1]
        ins, Wps = saved[2:2 + n], saved[2 + n:2 + 2 * n]
        g = d_out.contiguous()
        if ctx.final_sigmoid:
            g = g * out * (1.0 - out)
        Q = g.shape[0]
        grads = [None] * (2 * n)
        dx = torch.zeros(Q, kp, device=g.device) if ctx.x_needs else None
        needs = ctx.needs_input_grad[4:]
        items, item_layer = [], []
        for li in range(n - 1, -1, -1):
            inp, Wp = ins[li], Wps[li]
            if needs[2 * li]:  # dW (and db as its by-product): deferred, all layers go out as one grouped launch
                items.append(dict(A=g, B=inp, colsum=bool(needs[2 * li + 1])))
                item_layer.append(li)
            elif needs[2 * li + 1]:
                grads[2 * li + 1] = hip.colsum(g)
            if li > 0:
                if ctx.x_needs and li - 1 == ctx.skip_at:
                    dx += hip.gemm(g, Wp[:, width:], trans_a=False, trans_b=False)
                # d(prev hidden) = (g @ W[:, :width]) * relu'(prev)
                g = hip.gemm(g, Wp[:, :width], trans_a=False, trans_b=False, epi=hip.EPI_MUL_POS,
                             aux_in=inp[:, :width])
            elif ctx.x_needs:
                dx += hip.gemm(g, Wp, trans_a=False, trans_b=False)
        if items:
            for li, (dWp, db) in zip(item_layer, hip.gemm_tn_grouped(items)):
                if db is not None:
                    grads[2 * li + 1] = db
                if li == 0:
                    grads[0] = dWp[:, in_cols] if in_cols is not None else dWp
                elif li - 1 == ctx.skip_at:
                    grads[2 * li] = torch.cat([dWp[:, :width], dWp[:, width + in_cols]], dim=1)
                else:
                    grads[2 * li] = dWp
        return (dx, None, None, None) + tuple(grads)


def relu_mlp(x, in_cols, skip_at, final_sigmoid, weights, biases):
    params = []
    for W, b in zip(weights, biases):
        params += [W, b]
    return ReluMLP.apply(x, in_cols, skip_at, final_sigmoid, *params)


class FusedPairMLP(torch.autograd.Function):
    """256-wide ReLU MLP evaluated on the virtual rows [A[q % nA] | B[q // nA]] (light-major flattening of
    stage2/model/renderer.py:163,193) with the fully fused inference kernel.  Used for the L-light
    visibility branch, which the reference evaluates with gradients enabled but whose result only
    enters the loss detached (renderer.py:197, loss.py:82-83).  If a caller does backpropagate into it,
    backward re-evaluates the rows on the training path."""

    @staticmethod
    def forward(ctx, tab_a, tab_b, in_cols, skip_at, *params):
        _hit('FusedPairMLP')
        Ws, bs = params[0::2], params[1::2]
        nA, nB = tab_a.shape[0], tab_b.shape[0]
        din_half = in_cols.numel() // 2
        packed = fused.pack_relu_mlp(list(Ws), list(bs), din_half, din_half, skip_at)
        out = packed(tab_a, nA * nB, a_div=1, a_mod=nA, tab_b=tab_b, b_div=nA, b_mod=nB)
        ctx.save_for_backward(tab_a, tab_b, in_cols, *params)
        ctx.skip_at = skip_at
        return out

    @staticmethod
    def backward(ctx, d_out):
        tab_a, tab_b, in_cols = ctx.saved_tensors[:3]
        params = [p.detach().requires_grad_(True) for p in ctx.saved_tensors[3:]]
        nA, nB = tab_a.shape[0], tab_b.shape[0]
        ta = tab_a.detach().requires_grad_(ctx.needs_input_grad[0])
        tb = tab_b.detach().requires_grad_(ctx.needs_input_grad[1])
        with torch.enable_grad():
            x = torch.cat([ta.tile(nB, 1), tb.repeat_interleave(nA, dim=0)], dim=1)
            out = ReluMLP.apply(x, in_cols, ctx.skip_at, False, *params)
            wanted = [t for t in [ta, tb] + params if t.requires_grad]
            got = torch.autograd.grad(out, wanted, d_out, allow_unused=True)
        it = iter(got)
        res = []
        for t in [ta, tb]:
            res.append(next(it) if t.requires_grad else None)
        pg = [next(it) for _ in params]
        return (res[0], res[1], None, None) + tuple(pg)


# --------------------------------------------------------------------------- dense per-pixel outputs
class SplitRows(torch.autograd.Function):
    """(t[:k], t[k:]) whose backward is ONE concatenation.  Two plain slices cost two zero fills, two copies and an add in
    backward (five launch-bound kernels, on the stage-2 side stream each waits for a free CU beside the visibility chain)."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.k, ctx.shape = k, t.shape
        return t[:k], t[k:]

    @staticmethod
    def backward(ctx, ga, gb):
        k, shape = ctx.k, ctx.shape
        if ga is None:
            ga = gb.new_zeros((k,) + tuple(shape[1:]))
        if gb is None:
            gb = ga.new_zeros((shape[0] - k,) + tuple(shape[1:]))
        return torch.cat([ga, gb], dim=0), None


class ScatterRows(torch.autograd.Function):
    """All dense outputs of PSNetwork.forward in one launch: dense_k [B_k, N, C_k] = fill_k everywhere except the surface
    pixels idx, which carry rows_k [B_k*Ns, C_k] (light-major; stage2/model/renderer.py:145-152, 204-264).
    apply(idx [Ns] int64, inv [N] int32, specs ((B, C, fill), ...), *rows) -> tuple of dense tensors.  Backward gathers the
    rows back (one launch for all outputs that received a gradient)."""

    @staticmethod
    def forward(ctx, idx, inv, specs, *rows):
        _hit('ScatterRows')
        ctx.set_materialize_grads(False)  # dense outputs nobody differentiates: None, not a zero-filled [B, N, C] + its gather
        ctx.specs, ctx.n_pix, ctx.ns = specs, inv.numel(), idx.numel()
        ctx.save_for_backward(idx, inv)
        return tuple(hip.scatter_rows(list(specs), [r.detach() for r in rows], inv, inv.numel(), idx.numel()))

    @staticmethod
    def backward(ctx, *grads):
        idx, inv = ctx.saved_tensors
        sel = [k for k, g in enumerate(grads) if g is not None and ctx.needs_input_grad[3 + k]]
        out = [None] * len(grads)
        if sel:
            # (inv: the rows of a padded index list that are not the first of their pixel are dead and receive exact zeros)
            got = hip.gather_rows([ctx.specs[k] for k in sel], [grads[k].contiguous() for k in sel], idx, ctx.n_pix, ctx.ns, inv=inv)
            for k, g in zip(sel, got):
                out[k] = g
        return (None, None, None) + tuple(out)


# --------------------------------------------------------------------------- weight normalisation
class WeightNormAll(torch.autograd.Function):
    """Effective matrices of all weight-normalised layers of a network, w_l = v_l (g_l / |v_l|_row) * scale_l
    (nn.utils.weight_norm as in stage1/model/network.py:37-66), one launch forward and one backward
    (csrc/weight_norm.hip) instead of ~4 + ~12 torch kernels per layer.
    apply(scales (tuple of floats), g_0, v_0, g_1, v_1, ...) -> (w_0, w_1, ...)."""

    @staticmethod
    def forward(ctx, scales, *gv):
        _hit('WeightNormAll')
        gs, vs = [t.detach().contiguous() for t in gv[0::2]], [t.detach().contiguous() for t in gv[1::2]]
        ctx.scales = scales
        ctx.save_for_backward(*gs, *vs)
        return tuple(hip.weight_norm_fwd(vs, [g.reshape(-1) for g in gs], scales))

    @staticmethod
    def backward(ctx, *dws):
        n = len(ctx.scales)
        sv = ctx.saved_tensors
        gs, vs = sv[:n], sv[n:]
        dws = [torch.zeros_like(v) if d is None else d.contiguous() for d, v in zip(dws, vs)]
        dvs, dgs = hip.weight_norm_bwd(list(vs), [g.reshape(-1) for g in gs], ctx.scales, dws)
        out = [None]
        for g, dg, dv in zip(gs, dgs, dvs):
            out += [dg.view_as(g), dv]
        return tuple(out)


# --------------------------------------------------------------------------- SG shading
def sg_shade(light_dir, view, normal, albedo, weights, lobe, light_int, vis, specular_rgb):
    """Spherical-Gaussian shading over the light-major rows (l, n) -> l*Ns + n (csrc/shade.hip).

    stage2/model/sgbasis.py:16-32 + stage2/model/renderer.py:174-204:
        h = normalize(l + v);  D_k = exp(lambda_k (h.n - 1));  spec_c = max(sum_k w_{c,k} D_k, 0)
        rgb = clamp((albedo + spec) * I_l * (l.n) * clamp(vis, 0, 1), 0, 1)      (cos is NOT clamped)
    light_dir [L,3], view/normal/albedo [Ns,3], weights [Ns,nbasis], light_int [L,1] tensor or float,
    vis [L*Ns,1] or None.  Returns rgb [L*Ns,3], spec [L*Ns,3 or 1]."""
    return SGShade.apply(light_dir, view, normal, albedo, weights, lobe, light_int, vis, specular_rgb)


class SGShade(torch.autograd.Function):
    """Fused SG shading (csrc/shade.hip); backward recomputes the forward and returns gradients for the light
    directions / intensities, normals, albedo, SG weights and (if it was not detached) the visibility."""

    @staticmethod
    def forward(ctx, light_dir, view, normal, albedo, weights, lobe, light_int, vis, specular_rgb):
        _hit('SGShade')
        li_t, li_s = None, 0.0
        rgb_lights = False
        if torch.is_tensor(light_int):
            if light_int.dim() == 2 and light_int.shape[1] == 3:  # RGB environment lights (stage2/eval.py:200)
                if any(ctx.needs_input_grad):
                    raise RuntimeError('sg_shade: RGB light intensities are supported on the forward-only path')
                li_t = light_int.contiguous()
                rgb_lights = True
            elif light_int.numel() > 1:
                li_t = light_int.reshape(-1).contiguous()
            else:
                li_t = light_int.reshape(-1).expand(light_dir.shape[0]).contiguous()
        else:
            li_s = float(light_int)
        ctx.set_materialize_grads(False)
        vis_f = None if vis is None else vis.reshape(-1).contiguous()
        args = (light_dir.contiguous(), view.contiguous(), normal.contiguous(), albedo.contiguous(),
                weights.contiguous(), lobe.contiguous())
        rgb, spec = hip.sg_shade_fwd(*args, li_t, li_s, vis_f, specular_rgb)
        ctx.save_for_backward(*args, *([li_t] if li_t is not None else []), *([vis_f] if vis_f is not None else []))
        ctx.has_li, ctx.has_vis, ctx.li_s, ctx.specular_rgb = li_t is not None, vis_f is not None, li_s, specular_rgb
        ctx.li_shape = light_int.shape if torch.is_tensor(light_int) else None
        ctx.vis_shape = None if vis is None else vis.shape
        return rgb, spec

    @staticmethod
    def backward(ctx, g_rgb, g_spec):
        sv = list(ctx.saved_tensors)
        light_dir, view, normal, albedo, weights, lobe = sv[:6]
        k = 6
        li_t = sv[k] if ctx.has_li else None
        k += 1 if ctx.has_li else 0
        vis_f = sv[k] if ctx.has_vis else None
        want_vis = ctx.has_vis and ctx.needs_input_grad[7]
        g_spec_c = None if g_spec is None else g_spec.contiguous()
        if g_rgb is None:
            g_rgb = torch.zeros(light_dir.shape[0] * view.shape[0], 3, device=view.device)
        d_alb, d_w, d_n, d_vis, d_ld, d_li = hip.sg_shade_bwd(light_dir, view, normal, albedo, weights, lobe, li_t,
                                                              ctx.li_s, vis_f, ctx.specular_rgb, g_rgb.contiguous(),
                                                              g_spec_c, want_vis)
        if d_li is not None:
            d_li = d_li.sum().reshape(ctx.li_shape) if int(torch.tensor(ctx.li_shape).prod()) == 1 else d_li.reshape(ctx.li_shape)
        if d_vis is not None:
            d_vis = d_vis.reshape(ctx.vis_shape)
        return d_ld, None, d_n, d_alb, d_w, None, d_li, d_vis, None


# --------------------------------------------------------------------------- stage-1 geometry field
def _buf(rows, cols, device):
    """[rows, cols] view of a buffer whose row stride is a multiple of 4 floats (16-byte aligned rows)."""
    pad = (cols + 3) // 4 * 4
    return torch.empty(rows, pad, device=device, dtype=torch.float32)[:, :cols]


class GeoField(torch.autograd.Function):
    """Occupancy ("geo") MLP of stage1/model/network.py:85-95 together with its spatial gradient
    (network.py:108-120), in ONE differentiable op:

        out  [Q, F+1] = infer_occ(p)          (logit = out[:, 0], features = out[:, 1:])
        grad [Q, 3]   = d out[:, 0] / d p     (what autograd.grad(..., create_graph=True) returns)

    The reference obtains ``grad`` by running infer_occ a second time and calling autograd with
    create_graph=True, so training differentiates through a gradient (double backward through nine
    weight-normed softplus(beta=100) layers).  Here the reverse sweep that produces ``grad`` is written
    out as explicit GEMMs (r_l = (r_{l+1} * sigmoid(100 z_l)) W_l), and backward() is the hand-derived
    adjoint of BOTH passes -- no autograd-of-autograd, and infer_occ runs once instead of twice.

    Inputs: p [Q,3]; n_octaves; scale (= 1/rescale); skips (layers whose input is cat[x, pe]; the
    1/sqrt(2) is folded into those layers' weights by the caller); with_grad; then W_0, b_0, W_1, ...
    as EFFECTIVE dense weights (weight-norm is applied by the caller with torch ops on the tiny weight
    tensors, so autograd maps dW back to weight_g / weight_v).
    Points are never differentiated on the reference's training path (sample depths are detached,
    rendering.py:88-101), so no gradient is returned for p."""

    @staticmethod
    def forward(ctx, p, n_octaves, scale, skips, with_grad, *params):
        _hit('GeoField')
        Ws, bs = [w.contiguous() for w in params[0::2]], [b.contiguous() for b in params[1::2]]
        n = len(Ws)
        p = p.contiguous()
        Q = p.shape[0]
        dev = p.device
        d_pe = 3 + 6 * n_octaves
        kp = (d_pe + 3) // 4 * 4
        H = Ws[1].shape[1]
        pe = hip.pe_encode(p, n_octaves, kp, scale)
        W0p = torch.nn.functional.pad(Ws[0], (0, kp - d_pe)).contiguous()
        A, S = [None] * n, [None] * (n - 1)
        A[0] = pe
        for l in range(n - 1):
            o = Ws[l].shape[0]
            if (l + 1) in skips:
                nxt = torch.empty(Q, o + d_pe, device=dev, dtype=torch.float32)
                dst = nxt[:, :o]
                nxt[:, o:] = pe[:, :d_pe]
            else:
                nxt = _buf(Q, o, dev)
                dst = nxt
            S[l] = _buf(Q, o, dev)
            hip.gemm(A[l], W0p if l == 0 else Ws[l], trans_b=True, bias=bs[l], epi=hip.EPI_BIAS_SOFTPLUS, out=dst,
                     aux_out=S[l])
            A[l + 1] = nxt
        out = hip.gemm(A[n - 1], Ws[n - 1], trans_b=True, bias=bs[n - 1], epi=hip.EPI_BIAS)
        grad = None
        U, R = [None] * (n - 1), [None] * n
        if with_grad:
            w_last = Ws[n - 1][0:1, :]  # d logit / d a_{n-1}, the same row for every point
            U[n - 2] = S[n - 2] * w_last
            d_pe_acc = None
            for l in range(n - 2, 0, -1):
                in_a = Ws[l].shape[1] - d_pe if l in skips else Ws[l].shape[1]
                R[l] = _buf(Q, in_a, dev)
                U[l - 1] = _buf(Q, in_a, dev)
                hip.gemm(U[l], Ws[l][:, :in_a], epi=hip.EPI_MUL_AUX_RAW, aux_in=S[l - 1], out=U[l - 1], aux_out=R[l])
                if l in skips:
                    g_skip = hip.gemm(U[l], Ws[l][:, in_a:])
                    d_pe_acc = g_skip if d_pe_acc is None else d_pe_acc + g_skip
            d_pe_t = hip.gemm(U[0], W0p)  # [Q, kp]
            if d_pe_acc is not None:
                d_pe_t[:, :d_pe] += d_pe_acc
            grad = hip.pe_encode_bwd(p, d_pe_t, n_octaves, scale)
        if any(ctx.needs_input_grad):
            ctx.n, ctx.skips, ctx.with_grad, ctx.d_pe, ctx.kp = n, list(skips), with_grad, d_pe, kp
            ctx.n_octaves, ctx.scale = n_octaves, scale
            keep = [p, W0p] + Ws + A + S
            if with_grad:
                keep += U + R[1:n - 1]
            ctx.save_for_backward(*keep)
        if grad is None:
            grad = torch.zeros(Q, 3, device=dev)
            ctx.mark_non_differentiable(grad)
        return out, grad

    @staticmethod
    def backward(ctx, d_out, d_grad):
        n, skips, d_pe, kp = ctx.n, ctx.skips, ctx.d_pe, ctx.kp
        sv = list(ctx.saved_tensors)
        p, W0p = sv[0], sv[1]
        Ws = sv[2:2 + n]
        A = sv[2 + n:2 + 2 * n]
        S = sv[2 + 2 * n:2 + 3 * n - 1]
        Q = p.shape[0]
        dev = p.device
        dW = [None] * n
        db = [None] * n
        sweep = ctx.with_grad and d_grad is not None
        dS = [None] * (n - 1)

        def add_dW(l, a_t, b_mat, with_bias=False):  # dW[l] (+)= a_t^T @ b_mat ; db[l] = column sums of a_t
            sk = _split_k_for(Q, a_t.shape[1], b_mat.shape[1])
            cs = torch.empty(a_t.shape[1], device=dev) if with_bias else None
            if dW[l] is None:
                dW[l] = hip.gemm(a_t, b_mat, trans_a=True, split_k=sk, colsum_a=cs)
            else:
                hip.gemm(a_t, b_mat, trans_a=True, split_k=sk, out=dW[l], epi=hip.EPI_ACCUM, colsum_a=cs)
            if with_bias:
                db[l] = cs

        if sweep:
            base = 2 + 3 * n - 1
            U = sv[base:base + n - 1]
            R = [None] + sv[base + n - 1:base + n - 1 + (n - 2)] + [None]  # R[1..n-2]
            dd_pe = hip.pe_encode_jvp(p, d_grad.contiguous(), ctx.n_octaves, kp, ctx.scale)  # dL/d(d_pe)
            w_last = Ws[n - 1][0:1, :]
            dR = dd_pe  # dL/dR[0]
            for l in range(n - 1):
                Wl = W0p if l == 0 else Ws[l]
                o = Wl.shape[0]
                r_next = R[l + 1] if l + 1 <= n - 2 else w_last.expand(Q, -1)  # stride-0 rows for the last layer
                if (l + 1) in skips:  # the adjoint of R[l+1] is [a-part | pe-part]; build it in one buffer
                    nxt = torch.empty(Q, o + d_pe, device=dev, dtype=torch.float32)
                    dst = nxt[:, :o]
                    nxt[:, o:] = dd_pe[:, :d_pe]
                else:
                    nxt = _buf(Q, o, dev)
                    dst = nxt
                dS[l] = _buf(Q, o, dev)
                # du = dR @ W_l^T ;  dR[l+1](a-part) = du * S_l ;  dS_l = du * R[l+1]
                hip.gemm(dR, Wl, trans_b=True, epi=hip.EPI_MUL2, aux_in=S[l], aux_in2=r_next, out=dst, aux_out=dS[l])
                add_dW(l, U[l], dR)
                dR = nxt
            # R[n-1] is row 0 of the last layer broadcast over points
            dW[n - 1] = torch.zeros_like(Ws[n - 1])
            dW[n - 1][0] = hip.colsum(dR)

        g = d_out.contiguous()
        add_dW(n - 1, g, A[n - 1], with_bias=True)
        for l in range(n - 1, 0, -1):
            in_a = Ws[l].shape[1] - d_pe if l in skips else Ws[l].shape[1]
            g_prev = _buf(Q, in_a, dev)
            if sweep:
                hip.gemm(g, Ws[l][:, :in_a], epi=hip.EPI_SOFTPLUS_BWD, aux_in=S[l - 1], aux_in2=dS[l - 1], out=g_prev)
            else:
                hip.gemm(g, Ws[l][:, :in_a], epi=hip.EPI_MUL_AUX, aux_in=S[l - 1], out=g_prev)
            g = g_prev
            add_dW(l - 1, g, A[l - 1], with_bias=True)
        dW[0] = dW[0][:, :d_pe]
        grads = []
        for l in range(n):
            grads += [dW[l], db[l]]
        return (None, None, None, None, None) + tuple(grads)


# --------------------------------------------------------------------------- GGX microfacet shading
class MFShade(torch.autograd.Function):
    """train.render_model = microfacet: stage2/model/microfacet.py:35-114 + renderer.py:187-204, fused
    (csrc/shade.hip).  rough [Ns,1] is the sigmoid output of rough_net; returns rgb [L*Ns,3]."""

    @staticmethod
    def forward(ctx, light_dir, view, normal, albedo, rough, light_int, vis, f0):
        _hit('MFShade')
        li_t, li_s = None, 0.0
        if torch.is_tensor(light_int):
            li_t = light_int.reshape(-1).expand(light_dir.shape[0]).contiguous() if light_int.numel() == 1 \
                else light_int.reshape(-1).contiguous()
        else:
            li_s = float(light_int)
        vis_f = None if vis is None else vis.reshape(-1).contiguous()
        args = (light_dir.contiguous(), view.contiguous(), normal.contiguous(), albedo.contiguous(),
                rough.reshape(-1).contiguous())
        rgb = hip.mf_shade_fwd(*args, li_t, li_s, f0, vis_f)
        ctx.save_for_backward(*args, *([li_t] if li_t is not None else []), *([vis_f] if vis_f is not None else []))
        ctx.has_li, ctx.has_vis, ctx.li_s, ctx.f0 = li_t is not None, vis_f is not None, li_s, f0
        ctx.li_shape = light_int.shape if torch.is_tensor(light_int) else None
        ctx.vis_shape = None if vis is None else vis.shape
        ctx.rough_shape = rough.shape
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        sv = list(ctx.saved_tensors)
        light_dir, view, normal, albedo, rough = sv[:5]
        k = 5
        li_t = sv[k] if ctx.has_li else None
        k += 1 if ctx.has_li else 0
        vis_f = sv[k] if ctx.has_vis else None
        want_vis = ctx.has_vis and ctx.needs_input_grad[6]
        d_alb, d_r, d_n, d_vis, d_ld, d_li = hip.mf_shade_bwd(light_dir, view, normal, albedo, rough, li_t, ctx.li_s,
                                                              ctx.f0, vis_f, g_rgb.contiguous(), want_vis)
        if d_li is not None:
            n_li = 1
            for d in ctx.li_shape:
                n_li *= d
            d_li = d_li.sum().reshape(ctx.li_shape) if n_li == 1 else d_li.reshape(ctx.li_shape)
        if d_vis is not None:
            d_vis = d_vis.reshape(ctx.vis_shape)
        return d_ld, None, d_n, d_alb, d_r.reshape(ctx.rough_shape), d_li, d_vis, None


def mf_shade(light_dir, view, normal, albedo, rough, light_int, vis, f0):
    return MFShade.apply(light_dir, view, normal, albedo, rough, light_int, vis, f0)


# --------------------------------------------------------------------------- visibility net: shading + supervision rows
# ReLU-backward chains read the forward launch's SIGN BITS (32 bytes per row and layer, written beside the activation dumps:
# psn_mlp_infer_bits / PSN_ACT_RELU_BITS) instead of re-reading the 1 KB activation rows as masks -- d z = d h * (h > 0) either
# way, bit for bit; 16 operand loads per lane and layer fewer in the chain (each VMEM instruction of a stage costs the chain
# about 1 %).  False: the activation rows themselves (PSN_ACT_RELU_MASK; A/B and tests).
RELU_SIGN_BITS = True


def _sign_bits(rows, n, device):
    return [torch.empty(rows, 4, device=device, dtype=torch.int64) for _ in range(n)]


class VisibilityPair(torch.autograd.Function):
    """stage-2 visibility_net on BOTH row groups of a training step in one fused launch:
         rows [0, L*Ns)          shading lights  (renderer.py:191-200; enter the loss detached, :197)
         rows [L*Ns, (L+V)*Ns)   supervision lights 'vis_train' (renderer.py:251-262; trained by vis_loss)
    The supervision rows ride along with the gradient-free rows through the register-resident kernel, which
    dumps their hidden activations (row-major) for the backward pass; the backward is the usual ReLU chain on
    the fp32-MFMA GEMMs.  The layers that read the input block [PE(x_n) | PE(l_v)] use its separability:
    dW_x = (sum_v dz[v])^T PE(x), dW_l = (sum_n dz[:, n])^T PE(l) -- an 8x (resp. Ns x) shorter contraction.

    Inputs: pe_x [Ns,64], pe_l [L+V,64] (no gradient: points are data, light directions are detached by
    train.light_vis_detach), n_shade = L, in_cols, skip_at, then W0, b0, ...  Returns (vis [L*Ns,1], vis_t [V*Ns,1])."""

    @staticmethod
    def launch(pe_x, pe_l, n_shade, in_cols, skip_at, params, need_grad, packed=None, live_count=None):
        """The fused launch itself, separable from the autograd node: the renderer issues it at the very start of
        the forward pass (it is 60 % of the step and depends on nothing but points and lights) and attaches the
        node later with ``apply(..., pre, *params)``, so that the node keeps a LATE position in the graph and its
        (large) backward kernels are queued first, ahead of the many small launches of the other networks.
        live_count (float32 [1] on the device): pe_x is a surface list padded to a fixed capacity of which the first
        live_count[0] rows are real -- the workgroups of the gradient-free shading rows that hold padding only leave early
        (psn_mlp_infer_padded; their outputs are zeros, which nothing reads: ops.ScatterRows drops the padding rows)."""
        Ws, bs = params[0::2], params[1::2]
        Ns, LV = pe_x.shape[0], pe_l.shape[0]
        V = LV - n_shade
        n = len(Ws)
        din_half = in_cols.numel() // 2
        with torch.no_grad():
            if packed is None:
                packed = fused.pack_relu_mlp(list(Ws), list(bs), din_half, din_half, skip_at)
            save = [torch.empty(V * Ns, 256, device=pe_x.device) for _ in range(n - 1)] if (need_grad and V > 0) else None
            bits = _sign_bits(V * Ns, n - 1, pe_x.device) if (save is not None and RELU_SIGN_BITS) else None
            out = packed(pe_x, LV * Ns, a_div=1, a_mod=Ns, tab_b=pe_l, b_div=Ns, b_mod=LV, save=save, save_row0=n_shade * Ns,
                         live=None if (live_count is None or Ns % 64 != 0) else (live_count, Ns),  # (64-row blocks must not straddle groups)
                         save_bits=bits)
        return out, (save, bits) if save is not None else None

    @staticmethod
    def forward(ctx, pe_x, pe_l, n_shade, in_cols, skip_at, pre, *params):
        _hit('VisibilityPair')
        Ns, LV = pe_x.shape[0], pe_l.shape[0]
        V = LV - n_shade
        n = len(params) // 2
        need = any(ctx.needs_input_grad[6:])
        out, dumps = pre if pre is not None else VisibilityPair.launch(pe_x, pe_l, n_shade, in_cols, skip_at, params, need)
        save, bits = dumps if (need and dumps is not None) else (None, None)
        if save is not None:
            ctx.save_for_backward(pe_x, pe_l[n_shade:], in_cols, *save, *params)
        ctx.bits = bits  # (sign-bit words of the dumps, or None: plain tensors no gradient ever flows through)
        ctx.n, ctx.skip_at, ctx.V, ctx.saved = n, skip_at, V, save is not None
        # the standard column list [0 .. d) + [stride .. stride + d) (PSNetwork._cols): known without reading the device tensor
        ctx.contiguous_cols = bool(getattr(in_cols, '_psn_contiguous_pair', False))
        vis, vis_t = out[:n_shade * Ns], out[n_shade * Ns:]
        ctx.mark_non_differentiable(vis)
        return vis, vis_t

    @staticmethod
    def backward(ctx, _g_vis, g):
        n, V = ctx.n, ctx.V
        if not ctx.saved or g is None:
            return (None,) * (6 + 2 * n)
        sv = ctx.saved_tensors
        pe_x, pe_lv, in_cols = sv[0], sv[1], sv[2]
        H = sv[3:3 + n - 1]                  # post-ReLU outputs of layers 0..n-2
        Ws = sv[3 + n - 1::2]
        Ns = pe_x.shape[0]
        Q = V * Ns
        half = in_cols.numel() // 2
        cols_a, cols_b = in_cols[:half], in_cols[half:] - pe_x.shape[1]
        contiguous_cols = ctx.contiguous_cols
        grads = [None] * (2 * n)
        g = g.contiguous()

        # last layer (out = 1): dW = g^T h ; d h_{n-2} = g w  (rank-1, feeds the fused backward chain)
        # d z_l = d h_l * relu'(h_l), d h_{l-1} = W_l[:, :256]^T d z_l for l = n-2 .. 0 in ONE register-resident
        # launch (transposed weight packs, activations re-read as masks, every d z_l dumped for the weight GEMMs)
        bits = ctx.bits
        chain = fused.pack_relu_bwd(list(Ws), ctx.skip_at, bits=bits is not None, x3=CHAIN_X3)
        DZ = [torch.empty(Q, 256, device=g.device) for _ in range(n - 1)]  # DZ[j] = d z_{n-2-j}
        # (the rank-1 init table d h_{n-2} = g w is formed inside the chain kernel: hip.mlp_infer rank_init)
        chain(None, Q, a_div=1, a_mod=Q, rank_init=(g.reshape(Q, 1), Ws[n - 1].reshape(1, -1).contiguous()),
              mask=[(bits if bits is not None else H)[n - 2 - j] for j in range(n - 1)], save=DZ, save_row0=0)
        # Every weight gradient in one grouped launch.  The input block [PE(x_n) | PE(l_v)] of row k = v Ns + n is never
        # expanded: its two halves are TABLES read as pe_x[k % Ns] and pe_lv[k // Ns] by the GEMM itself, side by side in
        # one 128-column product.
        # last layer (one output): dW = g^T h as a weighted column sum, db = sum g -- a 1-row GEMM item would occupy a
        # whole 128-row tile per K slice
        grads[2 * (n - 1)] = hip.colsum(H[n - 2], row_weight=g)  # [1, 256]
        grads[2 * (n - 1) + 1] = _rowsum_small(g)
        items = []
        where = []
        xl = {}  # layer -> (d W_x [256, 64], d W_l [256, 64], bias gradient or None)
        for li in range(n - 2, -1, -1):
            dz = DZ[n - 2 - li]
            if li > 0:
                items.append(dict(A=dz, B=H[li - 1], colsum=True))
                where.append((li, 'w'))
            if li == 0 or li - 1 == ctx.skip_at:
                # The input block [PE(x_n) | PE(l_v)] of row k = v Ns + n is separable: sum the V light slices of dz first,
                #   d W_x = sum_n (sum_v dz[v, n])^T PE(x_n)   (K = Ns instead of V Ns),
                #   d W_l = sum_v (sum_n dz[v, n])^T PE(l_v)   (K = V),
                # two reductions that read dz once each instead of a K = V Ns product on 128-column tiles (which ran at
                # ~40 TF: the narrow-tile kernel is the slow one of the weight-gradient set).
                if V <= 16:
                    xl[li] = None  # all input layers together below: two launches (psn_pair_sums_group)
                else:
                    dz3 = dz.view(V, Ns, dz.shape[1])
                    dz_x, dz_l = dz3.sum(0), dz3.sum(1)
                    dWl = hip.gemm(dz_l, pe_lv.contiguous(), trans_a=True)          # [256, 64]
                    xl[li] = [dz_x, dWl, dz_l.sum(0) if li == 0 else None]
        grouped = [li for li in sorted(xl) if xl[li] is None]
        if grouped:
            # per layer: sum_v dz (the K = Ns operand of d W_x), d W_l = (sum_n dz[v])^T PE(l_v) and -- layer 0 -- the bias
            # gradient, every layer in the same two launches (was pair_sums + a reduction of its partials + a small GEMM each)
            pl = pe_lv.contiguous()
            for li, r in zip(grouped, hip.pair_sums_group([DZ[n - 2 - li] for li in grouped], V, Ns, pl, pl.shape[1],
                                                          [li == 0 for li in grouped])):
                xl[li] = list(r)
        # d W_x of both input layers (K = Ns, 256 x 64 outputs) in ONE launch of the 256 x 64-tile kernel + one reduction
        # (were a split-K GEMM + reduction each)
        lis = sorted(xl)
        for li, (C, _cs) in zip(lis, hip.gemm_tn_grouped([dict(A=xl[li][0], B=pe_x) for li in lis])):
            xl[li][0] = C
        parts = {}
        if items:
            for (li, kind), (C, cs) in zip(where, hip.gemm_tn_grouped(items)):
                parts[(li, kind)] = C
                if cs is not None:
                    grads[2 * li + 1] = cs
        for li in range(n - 1):
            blocks = [parts[(li, 'w')]] if (li, 'w') in parts else []
            if li in xl:  # columns: d W_x (table pe_x), d W_l (table pe_lv)
                dWx, dWl, cs0 = xl[li]
                # (the encoding columns of both tables are their leading columns: slices, not index kernels)
                blocks += [dWx[:, :half] if contiguous_cols else dWx[:, cols_a], dWl[:, :half] if contiguous_cols else dWl[:, cols_b]]
                if cs0 is not None:
                    grads[2 * li + 1] = cs0
            grads[2 * li] = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)
        return (None, None, None, None, None, None) + tuple(grads)


# --------------------------------------------------------------------------- stage-1 geometry field, fused chains
# One dump per softplus layer (round 5): the value pass writes A_l = softplus_100(z_l) only; the three consumer
# chains re-form sigmoid(100 z_l) = 1 - exp(-100 A_l) in their activation programs (PSN_ACT_*_A: a separate instantiation of the
# chain kernel -- as extra cases of the one kernel they spilled 92 registers).  Measured at 262,144 points (profiles/r05a_*single_dump*):
# F1 writes 2.42 GB instead of 4.58 GB and takes 6 % fewer cycles, F2 / B1 / B2 pay 1.2 - 2.3 % for the exponentials (+25 M vector
# instructions each); chains -0.3 % cycles, the configs[1] step -0.2 .. -0.45 ms (43.8 -> 43.5 ms), eight [Q, 256] tensors (4.3 GB at
# 524k render samples) are no longer allocated.  Results differ in the last bits (gradients 6e-7 relative).  PSN_GEO_SINGLE_DUMP=0
# restores the two-dump chains (A/B: tools/dbg/ab_single_dump.py).
import os as _os
GEO_SINGLE_DUMP = _os.environ.get('PSN_GEO_SINGLE_DUMP', '1') == '1'
# Split-bf16 form of the stage-1 chains (experiment, never the headline; BASELINE configs[4] "bf16 MFMA path"): the four geometry
# chains and the two appearance chains multiply on v_mfma_f32_16x16x32_bf16 -- every fp32 operand as two bf16 pieces, three partial
# products, fp32 accumulation (csrc/mlp_infer.hip stage_compute_x3; PsnMlpDesc.w_format = PSN_W_BF16X2) -- while the activation
# programs, dumps and epilogues stay fp32.  ``with ops.chain_precision('bf16x3'):`` around forward AND backward (the packs are
# rebuilt when the mode changes).
CHAIN_X3 = _os.environ.get('PSN_CHAIN_X3', '0') == '1'


class chain_precision(object):
    def __init__(self, mode):
        assert mode in ('fp32', 'bf16x3'), mode
        self.mode = mode

    def __enter__(self):
        global CHAIN_X3
        self.saved, CHAIN_X3 = CHAIN_X3, self.mode == 'bf16x3'
        return self

    def __exit__(self, *exc):
        global CHAIN_X3
        CHAIN_X3 = self.saved
        return False


class GeoFieldFused(torch.autograd.Function):
    """Same function as GeoField (occupancy logit, 256 features and d logit / d p of the stage-1 geometry MLP, with a
    hand-derived backward through both the value pass and the gradient sweep), but every layer-to-layer chain runs
    in the register-resident fused kernel (csrc/mlp_infer.hip, chain variant): four launches per call --
    value pass, reverse sweep, adjoint of the sweep, adjoint of the value pass -- whose per-layer dumps feed the
    split-K weight-gradient GEMMs.  The Linear / softplus / elementwise traffic of 34 GEMM launches never
    round-trips through HBM as GEMM operands.  256-wide networks with one skip layer only (fallback: GeoField).

    Returns (logit [Q,1], feat [Q,256], grad [Q,3]).  ``chains`` = fused.pack_geo_chains(...) (cached per step).
    ``feat_rows`` (None = Q): only the features of the first feat_rows rows are returned / receive a gradient -- the points
    behind them are evaluated for their gradient alone (the surface-normal points of rendering.py:200-212 riding behind
    the render samples: one set of launches instead of two, and no [Q,256] zero-padded d feat in backward)."""

    @staticmethod
    def forward(ctx, p, n_octaves, scale, skips, with_grad, chains, feat_rows, *params):
        _hit('GeoFieldFused')
        Ws = [w for w in params[0::2]]
        n = len(Ws)
        p = p.contiguous()
        Q, dev = p.shape[0], p.device
        feat_rows = Q if feat_rows is None else int(feat_rows)  # rows whose features are returned (and receive a gradient)
        assert 1 <= feat_rows <= Q
        d_pe = 3 + 6 * n_octaves
        d_a = chains['d_a']
        sk = skips[0]
        pe = hip.pe_encode(p, n_octaves, 64, scale)  # [Q,64] xin table (39 real columns)
        A = [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]   # A[l] = softplus output of layer l = input of l+1
        single = bool(chains.get('single_dump'))
        # single-dump experiment (GEO_SINGLE_DUMP): the sigmoids are not dumped, the consumer chains re-form them from A
        S = A if single else [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]
        feat = torch.empty(Q, 256, device=dev)
        logit = chains['fwd'](pe, Q, save=A + [feat], save2=None if single else S + [None, None])
        A[sk - 1][:, d_a:] = pe[:, :d_pe]  # the skip layer's input is [a (217) | pe (39)]: complete the dumped tile
        grad = None
        U = R = None
        if with_grad:
            U = [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]   # U[l] = R[l+1] * S[l]
            # Raw sweep values R[l] are NOT dumped (the adjoint chain works from U, see backward) -- except at the skip
            # layer, whose last d_pe columns are the sweep's contribution to d logit / d pe through the skip input.
            r_sk = torch.empty(Q, 256, device=dev)
            w_row = Ws[n - 1][0:1, :].contiguous()
            # of r_sk only the d_pe columns behind the skip layer's activations are read: the chain writes those 16-column
            # tiles only; r0 = u_0 W_0 [Q, d_pe] is the chain's final layer (64 outputs wide, 4 output tiles)
            t_sk = sum(1 << t for t in range(d_a // 16, (d_a + d_pe - 1) // 16 + 1))
            r0 = chains['sweep'](None, Q, a_div=1, a_mod=1, init_a_direct=w_row,
                                 mask=[S[n - 2 - j] for j in range(n - 1)] + [None],
                                 save=[U[n - 2 - j] for j in range(n - 1)],
                                 save2=[None] + [r_sk if n - 1 - j == sk else None for j in range(1, n - 1)] + [None],
                                 save2_tiles=[None] + [t_sk if n - 1 - j == sk else None for j in range(1, n - 1)] + [None])
            # d logit / d pe = r0 + the skip layer's columns of r_sk: both read in place
            grad = hip.pe_encode_bwd(p, r0, n_octaves, scale, add=r_sk[:, d_a:d_a + d_pe])
        if any(ctx.needs_input_grad):
            ctx.meta = (n, sk, d_pe, d_a, n_octaves, scale, with_grad, chains, feat_rows)
            keep = [p, pe] + Ws + A + ([] if single else S)
            if with_grad:
                keep += U
            ctx.save_for_backward(*keep)
        if grad is None:
            grad = torch.zeros(Q, 3, device=dev)
            ctx.mark_non_differentiable(grad)
        return logit, (feat if feat_rows == Q else feat[:feat_rows]), grad

    @staticmethod
    def backward(ctx, d_logit, d_feat, d_grad):
        n, sk, d_pe, d_a, n_octaves, scale, with_grad, chains, feat_rows = ctx.meta
        sv = list(ctx.saved_tensors)
        p, pe = sv[0], sv[1]
        Ws = sv[2:2 + n]
        A = sv[2 + n:2 + 2 * n - 1]
        single = bool(chains.get('single_dump'))
        S = A if single else sv[2 + 2 * n - 1:2 + 3 * n - 2]
        Q, dev = p.shape[0], p.device
        sweep = with_grad and d_grad is not None
        d_logit = torch.zeros(Q, 1, device=dev) if d_logit is None else d_logit.contiguous()
        d_feat = torch.zeros(feat_rows, 256, device=dev) if d_feat is None else d_feat.contiguous()  # [feat_rows, 256]
        w_row = Ws[n - 1][0:1, :].contiguous()
        dW = [None] * n
        db = [None] * n

        E = None
        if sweep:
            base = 2 + (2 if single else 3) * n - (1 if single else 2)
            U = sv[base:base + n - 1]
            dd_pe = hip.pe_encode_jvp(p, d_grad.contiguous(), n_octaves, 64, scale)  # adjoint of d_pe, [Q,64]
            dR = [dd_pe] + [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]   # dR[l], l = 0..n-1
            # Adjoint of the sweep u_l = r_{l+1} * s_l, r_l = W_l^T u_l:  du_l = W_l dr_l,  dr_{l+1} = du_l * s_l  and
            # ds_l = du_l * r_{l+1}.  ds_l only ever enters the value adjoint as s_l' ds_l = 100 s_l (1 - s_l) du_l r_{l+1}
            # = 100 (1 - s_l) (du_l * u_l): the chain therefore multiplies with the dumped U_l instead of a dumped R_{l+1}
            # (one tensor less written by the sweep and none read back), and dumps E_l = du_l * u_l.
            E = [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]
            chains['sweep_bwd'](dd_pe, Q, mask=list(S), aux2=list(U), save=dR[1:], save2=E)
            dR[sk][:, d_a:] = dd_pe[:, :d_pe]  # adjoint of the skip layer's [a | pe] sweep value

        # adjoint of the value pass: dz_l = s_l da_l + 100 (1 - s_l) E_l
        dZ = [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]  # dZ[l] = d loss / d z_l
        key = 'value_bwd' if sweep else 'value_bwd_nosweep'
        # the rank-1 term d_logit (x) w_row of W_last^T d_out is formed inside the chain kernel (rank_init)
        chains[key](None, Q, a_div=1, a_mod=Q, rank_init=(d_logit.reshape(Q, 1).contiguous(), w_row), act_init=d_feat, act_init_rows=feat_rows,
                    mask=[S[n - 2 - j] for j in range(n - 1)],
                    aux2=[E[n - 2 - j] for j in range(n - 1)] if sweep else None,
                    save=[dZ[n - 2 - j] for j in range(n - 1)])
        # Every weight gradient of the call in ONE grouped launch: dW_l = dZ_l^T A_{l-1} (+ U_l^T dR_l from the sweep),
        # the bias gradients are the column sums of dZ_l, a by-product of staging the A tiles.
        a_last = A[n - 2]
        items = []
        for l in range(n - 1):
            o, i_w = Ws[l].shape[0], Ws[l].shape[1]
            it = dict(A=dZ[l][:, :o], B=pe[:, :d_pe] if l == 0 else A[l - 1][:, :i_w], colsum=True)
            if sweep:
                it['A2'], it['B2'] = U[l][:, :o], dd_pe[:, :d_pe] if l == 0 else dR[l][:, :i_w]
            items.append(it)
        items.append(dict(A=d_feat, B=a_last[:feat_rows], colsum=True))  # the rows behind feat_rows have no d feat
        res = hip.gemm_tn_grouped(items)
        for l in range(n - 1):
            dW[l], db[l] = res[l]
        row0 = hip.colsum(a_last, row_weight=d_logit).reshape(-1)  # d_logit^T a_last without the [Q,256] product
        if sweep:
            row0 = row0 + hip.colsum(dR[n - 1])
        dW[n - 1] = torch.cat([row0.unsqueeze(0), res[n - 1][0]], dim=0)
        db[n - 1] = torch.cat([_rowsum_small(d_logit), res[n - 1][1]])
        grads = []
        for l in range(n):
            grads += [dW[l], db[l]]
        return (None, None, None, None, None, None, None) + tuple(grads)


# --------------------------------------------------------------------------- stage-1 appearance network, fused chains
class AppNetFused(torch.autograd.Function):
    """stage1 appearance MLP (network.py:98-106) on cat[x, features] without materialising the 289-wide input:
    the 256 geometry features are the chain's initial activations, x = [point, view encoding, normal] rides in the
    64-wide input-feature table.  One forward launch (hidden activations dumped) and one backward launch (ReLU
    chain over transposed packs, HEAD = d features), then one grouped weight-gradient launch.
    Inputs: x [Q,64] (d_x real columns, no gradient), normal [Q,3] (= columns d_x-3.. of x; receives a gradient),
    feat [Q,256].  Returns the pre-activation colour [Q,3]."""

    @staticmethod
    def forward(ctx, x, normal, feat, d_x, chains, *params):
        _hit('AppNetFused')
        n = len(params) // 2
        Q = x.shape[0]
        feat = feat.contiguous()
        need = any(ctx.needs_input_grad)
        H = [torch.empty(Q, 256, device=x.device) for _ in range(n - 1)] if need else None
        out = chains['fwd'](x, Q, act_init=feat, save=H)
        if need:
            ctx.save_for_backward(x, feat, *H, *params[0::2])
            ctx.meta = (n, d_x, chains)
        return out

    @staticmethod
    def backward(ctx, g):
        n, d_x, chains = ctx.meta
        sv = ctx.saved_tensors
        x, feat = sv[0], sv[1]
        H = sv[2:2 + n - 1]
        Ws = sv[2 + n - 1:]
        Q, dev = x.shape[0], x.device
        g = g.contiguous()
        DZ = [torch.empty(Q, 256, device=dev) for _ in range(n - 1)]  # DZ[j] = d z_{n-2-j}
        d_feat = torch.empty(Q, 256, device=dev)
        # d h_{n-2} = g W_last (3 colours): a rank-3 init formed inside the chain kernel instead of a K = 3 GEMM that
        # writes [Q, 256] to HBM for the chain to read back
        d_normal = chains['bwd'](None, Q, a_div=1, a_mod=Q, rank_init=(g, Ws[n - 1].contiguous()),
                                 mask=[H[n - 2 - j] for j in range(n - 1)] + [None, None], save=DZ + [d_feat])  # [Q,3] = d z_0 W_0[:, normal columns]
        dz0 = DZ[n - 2]
        items = [dict(A=dz0, B=x[:, :d_x], colsum=True), dict(A=dz0, B=feat)]
        items += [dict(A=DZ[n - 2 - l], B=H[l - 1], colsum=True) for l in range(1, n - 1)]
        res = hip.gemm_tn_grouped(items)
        grads = [torch.cat([res[0][0], res[1][0]], dim=1), res[0][1]]
        for l in range(1, n - 1):
            grads += [res[l + 1][0], res[l + 1][1]]
        # output layer (3 colours): g^T h as a weighted column sum (a 3-row GEMM item occupies a whole 128-row tile)
        grads += [hip.colsum(H[n - 2], row_weight=g), _rowsum_small(g)]
        return (None, d_normal, d_feat, None, None) + tuple(grads)


# --------------------------------------------------------------------------- ReLU MLP on encoded points, fused
class FusedReluNet(torch.autograd.Function):
    """stage-2 Network / Normal_Network (stage2/model/renderer.py:17-49) of width 128 or 256 on a table of encoded
    points pe [Q, 64] (din real columns first): ONE register-resident forward launch that leaves the hidden
    activations behind, ONE backward chain launch (ReLU masks over transposed packs) and one grouped weight-gradient
    launch -- instead of one GEMM launch per layer and direction on Ns rows, which are all latency-bound.
    The points are data (no gradient with respect to pe)."""

    @staticmethod
    def pack(Ws, bs, din, skip_at, final_sigmoid, width):
        """Forward weight pack; callers may build it ahead of time (it only depends on the parameters)."""
        with torch.no_grad():
            return fused.pack_relu_mlp(list(Ws), list(bs), din, 0, skip_at,
                                       out_act=hip.OUT_SIGMOID if final_sigmoid else hip.OUT_NONE, precompute=False, width=width)

    @staticmethod
    def forward(ctx, pe, din, skip_at, final_sigmoid, width, packed, *params):
        _hit('FusedReluNet')
        Ws, bs = params[0::2], params[1::2]
        n = len(Ws)
        Q = pe.shape[0]
        if packed is None:
            packed = FusedReluNet.pack(Ws, bs, din, skip_at, final_sigmoid, width)
        need = any(ctx.needs_input_grad[6:])
        H = [torch.empty(Q, width, device=pe.device) for _ in range(n - 1)] if need else None
        bits = _sign_bits(Q, n - 1, pe.device) if (need and RELU_SIGN_BITS) else None
        out = packed(pe, Q, save=H, save_bits=bits)
        if need:
            ctx.save_for_backward(pe, out, *H, *Ws)
            ctx.meta = (n, din, skip_at, final_sigmoid, width)
            ctx.bits = bits
        return out

    @staticmethod
    def backward(ctx, g):
        n, din, skip_at, final_sigmoid, width = ctx.meta
        sv = ctx.saved_tensors
        pe, out = sv[0], sv[1]
        H = sv[2:2 + n - 1]
        Ws = sv[2 + n - 1:]
        Q, dev = pe.shape[0], pe.device
        g = g.contiguous()
        if final_sigmoid:
            g = torch.ops.aten.sigmoid_backward(g, out)  # g * (1 - out) * out, one launch
        bits = ctx.bits
        chain = fused.pack_relu_bwd(list(Ws), skip_at, width=width, bits=bits is not None, x3=CHAIN_X3)
        DZ = [torch.empty(Q, width, device=dev) for _ in range(n - 1)]  # DZ[j] = d z_{n-2-j}
        masks = [(bits if bits is not None else H)[n - 2 - j] for j in range(n - 1)]
        if Ws[n - 1].shape[0] <= 4:  # d h_{n-2} = g W_last as a rank-k init inside the kernel (albedo / normal nets: 3 outputs)
            chain(None, Q, a_div=1, a_mod=Q, rank_init=(g, Ws[n - 1].contiguous()), mask=masks, save=DZ)
        else:
            dh = hip.gemm(g, Ws[n - 1].contiguous())  # [Q, width]
            chain(None, Q, a_div=1, a_mod=Q, init_a_direct=dh, mask=masks, save=DZ)
        x_in = pe[:, :din]
        items = [dict(A=g, B=H[n - 2], colsum=True)]
        where = [(n - 1, 'w')]
        for l in range(n - 2, -1, -1):
            dz = DZ[n - 2 - l]
            if l == 0:
                items.append(dict(A=dz, B=x_in, colsum=True))
                where.append((0, 'w'))
            else:
                items.append(dict(A=dz, B=H[l - 1], colsum=True))
                where.append((l, 'w'))
                if l - 1 == skip_at:
                    items.append(dict(A=dz, B=x_in))
                    where.append((l, 'x'))
        res = hip.gemm_tn_grouped(items)
        dW, dWx, db = [None] * n, [None] * n, [None] * n
        for (l, kind), (C, cs) in zip(where, res):
            if kind == 'w':
                dW[l], db[l] = C, cs
            else:
                dWx[l] = C
        grads = []
        for l in range(n):
            grads += [dW[l] if dWx[l] is None else torch.cat([dW[l], dWx[l]], dim=1), db[l]]
        return (None, None, None, None, None, None) + tuple(grads)


# --------------------------------------------------------------------------- stage-2 losses, fused
class Stage2Losses(torch.autograd.Function):
    """MainLoss + NormalLoss of stage 2 (stage2/model/loss.py:27-92,123-141) over the dense model outputs in two launches
    forward and one backward (csrc/loss.hip) instead of ~75 elementwise / reduction launches.
    apply(rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j, mask_a, mask_b, l2, inv_denom, weight[, count_dev])
    -> (total (0-dim), terms [6] (no gradient)).  Tensors of inactive terms are None; inv_denom / weight: 6 floats;
    count_dev: optional device float [1] = masked-pixel count (then inv_denom excludes it, see psn_stage2_loss_fwd)."""

    @staticmethod
    def forward(ctx, rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j, mask_a, mask_b, l2, inv_denom, weight,
                count_dev=None):
        _hit('Stage2Losses')
        c = lambda t: None if t is None else t.detach().contiguous()
        ts = [c(t) for t in (rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j)]
        ma, mb = mask_a.contiguous(), mask_b.contiguous()
        out = hip.stage2_loss_fwd(*ts, ma, mb, l2, inv_denom, weight, count_dev)
        # save_for_backward (not plain ctx attributes): autograd then detects an in-place edit of any input between forward and
        # backward and the tensors go through the saved-tensor hooks; None slots are remembered by position
        slots = ts + [ma, mb, count_dev]
        ctx.present = [t is not None for t in slots]
        ctx.save_for_backward(*[t for t in slots if t is not None])
        ctx.l2 = l2
        ctx.k = [w * d for w, d in zip(weight, inv_denom)]
        ctx.need = ctx.needs_input_grad
        terms = out[:6]
        ctx.mark_non_differentiable(terms)
        return out[6], terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        saved = iter(ctx.saved_tensors)
        slots = [next(saved) if here else None for here in ctx.present]
        rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j, mask_a, mask_b, count_dev = slots
        k, ng = ctx.k, ctx.need
        need = set()
        if rgb is not None and ng[0] and k[0] != 0.0: need.add('rgb')
        if alb is not None and (ng[2] or ng[3]) and k[1] != 0.0: need.add('alb')
        if wgt is not None and (ng[4] or ng[5]) and k[2] != 0.0: need.add('wgt')
        if vis is not None and ng[6] and k[3] != 0.0: need.add('vis')
        if nrm is not None and (ng[8] or ng[10]) and (k[4] != 0.0 or k[5] != 0.0): need.add('nrm')
        d = {}
        if need:
            d = hip.stage2_loss_bwd(g_total.reshape(1).contiguous(), rgb, rgb_gt, k[0], alb, alb_j, k[1], wgt, wgt_j, k[2], vis, vis_gt,
                                    k[3], nrm, nrm_gt, nrm_j, k[4], k[5], mask_a, mask_b, ctx.l2, need, count_dev)
        g = d.get
        return (g('rgb'), None, g('alb'), g('alb_j'), g('wgt'), g('wgt_j'), g('vis'), None, g('nrm'), None, g('nrm_j'),
                None, None, None, None, None, None)


class SurfaceNormals(torch.autograd.Function):
    """stage1/model/rendering.py:200-212 in one launch each way (csrc/loss1.hip): apply(g [2 N, 3], hit [N] bool) ->
    (normal_pred [N, 3], diff_norm [N])."""

    @staticmethod
    def forward(ctx, g, hit):
        _hit('SurfaceNormals')
        gc, hc = g.detach().contiguous(), hit.contiguous()
        ctx.save_for_backward(gc, hc)
        return hip.surface_normals_fwd(gc, hc)

    @staticmethod
    def backward(ctx, d_norm_pred, d_diff):
        g, hit = ctx.saved_tensors
        c = lambda t: None if t is None else t.contiguous()
        return hip.surface_normals_bwd(g, hit, c(d_norm_pred), c(d_diff)), None


class Stage1Losses(torch.autograd.Function):
    """stage1/model/losses.py:24-70 over the outputs of the sync-free training forward (csrc/loss1.hip): two launches
    forward, one backward.  apply(rgb [N, 3], rgb_gt, diff [N] | None, hit [N] bool | None, normal [N, 3] | None, normal_gt,
    norm_mask [N] bool, acc [N] | None, mask_gt [N], mask_valid [N] bool, n_rays, weights (full, grad, norm, mask),
    reduce_counts) -> (loss (0-dim), terms [5] = colour / smoothness / normal / mask term and the total; no gradient).
    ``reduce_counts``: None, or callable(tensor [3]) summing the hit / norm_mask / mask_valid counts over ranks in place."""

    @staticmethod
    def forward(ctx, rgb, rgb_gt, diff, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, n_rays, weights, reduce_counts=None):
        _hit('Stage1Losses')
        c = lambda t: None if t is None else t.detach().contiguous()
        rgb, rgb_gt, diff, normal, normal_gt, acc, mask_gt = (c(t) for t in (rgb, rgb_gt, diff, normal, normal_gt, acc, mask_gt))
        hit, norm_mask, mask_valid = (c(t) for t in (hit, norm_mask, mask_valid))
        sums, terms = hip.stage1_loss_fwd(rgb, rgb_gt, diff, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, n_rays, weights,
                                          finish=reduce_counts is None)
        if reduce_counts is not None:
            reduce_counts(sums[4:7])
            terms = hip.stage1_loss_terms(sums, n_rays, weights, diff is not None and weights[1] != 0.0, normal is not None, acc is not None)
        slots = [rgb, rgb_gt, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, sums]
        ctx.present = [t is not None for t in slots]
        ctx.save_for_backward(*[t for t in slots if t is not None])
        ctx.n_rays, ctx.weights, ctx.has_diff = n_rays, tuple(float(w) for w in weights), diff is not None
        ctx.need = ctx.needs_input_grad
        loss = terms[4]
        ctx.mark_non_differentiable(terms)
        return loss, terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        saved = iter(ctx.saved_tensors)
        rgb, rgb_gt, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, sums = (next(saved) if here else None for here in ctx.present)
        w, ng = ctx.weights, ctx.need
        need = set()
        if ng[0] and w[0] != 0.0: need.add('rgb')
        if ctx.has_diff and ng[2] and w[1] != 0.0: need.add('diff')
        if normal is not None and ng[4] and w[2] != 0.0: need.add('normal')
        if acc is not None and ng[7] and w[3] != 0.0: need.add('acc')
        d = {}
        if need:
            d = hip.stage1_loss_bwd(g_loss.reshape(1).contiguous(), sums, rgb, rgb_gt, hit, normal, normal_gt, norm_mask, acc, mask_gt,
                                    mask_valid, ctx.n_rays, w, need)
        g = d.get
        return g('rgb'), None, g('diff'), None, g('normal'), None, None, g('acc'), None, None, None, None, None


# --------------------------------------------------------------------------- launch-bound row ops, fused (csrc/small.hip)
class NormalizeRows(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) for [n, 3] rows: one launch forward (norm, clamp_min, div), one backward (autograd's
    chain through div / clamp_min / norm is 14 launches)."""

    @staticmethod
    def forward(ctx, x, eps=1e-12):
        _hit('NormalizeRows')
        xc = x.detach().contiguous()
        ctx.save_for_backward(xc)
        ctx.eps = float(eps)
        return hip.normalize_rows_fwd(xc, eps)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return hip.normalize_rows_bwd(x, g.contiguous(), ctx.eps), None


def normalize_rows(x, eps=1e-12):
    """F.normalize(x, dim=-1); the fused form for fp32 [n, 3] device tensors, the torch formulation otherwise."""
    if x.is_cuda and x.dim() == 2 and x.shape[1] == 3 and x.dtype == torch.float32:
        return NormalizeRows.apply(x, eps)
    return torch.nn.functional.normalize(x, p=2, dim=-1, eps=eps)


class LightRows(torch.autograd.Function):
    """The light-table lookups of a stage-2 step (stage2/trainer.py:376-379): F.normalize(dir_table[idx]) and int_table[idx]
    in one launch; backward = the DENSE table gradients (as nn.Embedding(sparse=False) produces them) in one launch, without
    zero fills.  apply(dir_table [n, 3], int_table [n, 1], idx [L] int64) -> (dir [L, 3], inten [L, 1])."""

    @staticmethod
    def forward(ctx, dir_table, int_table, idx):
        _hit('LightRows')
        dt, it, ix = dir_table.detach().contiguous(), int_table.detach().contiguous(), idx.contiguous()
        ctx.save_for_backward(dt, ix)
        d, i = hip.light_rows_fwd(dt, it, ix)
        return d, i

    @staticmethod
    def backward(ctx, g_dir, g_int):
        dt, ix = ctx.saved_tensors
        need_d, need_i = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dd, di = hip.light_rows_bwd(dt, ix, g_dir.contiguous() if (need_d and g_dir is not None) else None,
                                    g_int.contiguous() if (need_i and g_int is not None) else None)
        return dd, di, None
