"""Host-side packing for the fused 256-wide MLP inference kernel (csrc/mlp_infer.hip).

A network is a list of layers, each reading the 256 activations of the previous
layer and/or the per-row INPUT FEATURES (positional encodings).  The input
block can be consumed in two ways:

* as MFMA k-tiles (``w_in``): up to four 32-float tiles, fetched into registers
  by each layer that consumes them, after its activation k-tiles (used by the
  stage-1 networks, whose query points are all distinct);
* as precomputed partial products (``init``): because the layer is linear in
  its input block, W_in [A_row | B_row] = W_a A_row + W_b B_row is evaluated
  ONCE per table row by two small GEMMs and the kernel starts the layer's
  accumulators from U[a] + V[b].  For the stage-2 visibility net the rows are
  all (surface point, light) pairs, so this removes layer 0 and the input half
  of the skip layer from the per-pair work (12 % of the MFMAs).

Dense weights are re-ordered once per optimiser step into MFMA fragment order
by ``psn_mlp_pack_layer``; this module only assembles the zero-padded dense
matrices with torch ops (tiny tensors) and fills the ``PsnMlpDesc``.
"""
import math

import torch

from . import hip


class PackedMLP(object):
    def __init__(self, desc, w, b, init_wa=None, init_wb=None, init_bias=None):
        self.desc, self.w, self.b = desc, w, b
        # stacked input-block weights of the layers evaluated through init tables
        self.init_wa, self.init_wb, self.init_bias = init_wa, init_wb, init_bias
        self.macs_per_row = None  # algorithmic MACs per row inside the kernel (true, unpadded block shapes)

    def on_points(self, points, pe_octaves, pe_scale, out=None, n_rows_dev=None, out_rows=None):
        """The network on gamma(pe_scale * points), [Q, 3] -> [Q, n_out], the encoding formed in the kernel prologue
        (hip.mlp_infer_pe): for packs whose input block is one positional encoding and that use no init tables.
        n_rows_dev / out_rows: a compacted list with a device-resident length, outputs scattered (see hip.mlp_infer_pe)."""
        return hip.mlp_infer_pe(self.desc, self.w, self.b, points, pe_octaves, pe_scale, out=out, macs_per_row=self.macs_per_row,
                                n_rows_dev=n_rows_dev, out_rows=out_rows)

    def __call__(self, tab_a, n_rows, a_div=1, a_mod=None, tab_b=None, b_div=1, b_mod=1, out=None, save=None,
                 save_row0=0, mask=None, init_a_direct=None, aux2=None, save2=None, act_init=None, rank_init=None,
                 save_tiles=None, save2_tiles=None, act_init_rows=None, live=None, save_bits=None):
        if a_mod is None:
            a_mod = tab_a.shape[0] if tab_a is not None else n_rows
        init_a = init_b = None
        if rank_init is not None:
            pass  # the init table as a rank-k product formed inside the kernel (hip.mlp_infer rank_init)
        elif init_a_direct is not None:
            init_a = init_a_direct  # caller-supplied init table (backward chains: d h of the last hidden layer)
        elif self.init_wa is not None:
            # U = A W_a^T (+ bias when there is no B table), V = B W_b^T + bias
            if self.init_wb is None:
                init_a = hip.gemm(tab_a, self.init_wa, trans_b=True, bias=self.init_bias, epi=hip.EPI_BIAS)
            else:
                init_a = hip.gemm(tab_a, self.init_wa, trans_b=True)
                init_b = hip.gemm(tab_b, self.init_wb, trans_b=True, bias=self.init_bias, epi=hip.EPI_BIAS)
        uses_in = any(self.desc.layers[i].n_kt_in > 0 for i in range(self.desc.n_layers))
        return hip.mlp_infer(self.desc, self.w, self.b, tab_a if uses_in else None, a_div, a_mod,
                             tab_b if uses_in else None, b_div, b_mod, n_rows, out=out, init_a=init_a, init_b=init_b,
                             save=save, save_row0=save_row0, mask=mask, aux2=aux2, save2=save2, act_init=act_init,
                             macs_per_row=self.macs_per_row, rank_init=rank_init, save_tiles=save_tiles, save2_tiles=save2_tiles,
                             act_init_rows=act_init_rows, live=live, save_bits=save_bits)


def _pad_cols(w, n):
    return torch.nn.functional.pad(w, (0, n - w.shape[1])) if w.shape[1] < n else w


def _pad_rows(w, n):  # F.pad with nothing to pad still clones: one fill + one copy kernel per call, ~50 per train step
    return torch.nn.functional.pad(w, (0, 0, 0, n - w.shape[0])) if w.shape[0] < n else w


def _pad_vec(b, n):
    return torch.nn.functional.pad(b, (0, n - b.shape[0])) if b.shape[0] < n else b


_ZEROS = {}


def _zeros(shape, device):
    """Cached constant zeros (never written): the backward chains are re-packed every step and would otherwise pay a
    fill launch per dummy bias / init block."""
    key = (tuple(shape) if not isinstance(shape, int) else (shape,), str(device))
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(*key[0], device=device)
        if z.is_cuda:
            # shared by every stream from now on (the stage-2 renderer packs on two): the fill must have happened before any of
            # them reads it -- one host wait per distinct size, ever
            torch.cuda.current_stream(z.device).synchronize()
    return z


TRIM_ACT_KTILES = True  # drop the last activation k-tile of a layer whose input has <= 32 * (n_mt - 1) columns (A/B switch)
DIRECT_INIT = 'direct'  # layer spec init_a=DIRECT_INIT: the init table is supplied by the caller at call time (no weights)


class Transposed(object):
    """Marker for pack_layers: use the transpose of ``w`` (backward chains) without materialising it."""

    def __init__(self, w):
        self.w = w.detach()


def _src(w):
    """(matrix, transpose flag, logical rows, logical cols) of a weight block handed to pack_layers."""
    if isinstance(w, Transposed):
        m = w.w
        return m, True, m.shape[1], m.shape[0]
    m = w.detach()
    return m, False, m.shape[0], m.shape[1]


def pack_layers(layers, in_kt_a, in_kt_b, n_out, out_act, device, has_final=True, width=256, reuse=None, x3=False):
    """layers: list of dicts {w_in | init_a/init_b, w_act, bias, act}.
    w_in: [o, <=in_kt*32] consumed as MFMA k-tiles; init_a [o, in_kt_a*32] / init_b [o, in_kt_b*32]: the same
    block evaluated through precomputed tables instead.  Hidden layers have o <= width (zero padded; width = 256 or
    128, one value per network), the final layer o = n_out <= 32 (<= 64 in a 256-wide chain).  w_act / w_in may be any row-major view with unit
    column stride (column slices of a parameter) or Transposed(w): the pack kernel reads them in place and zero-fills
    the padding, so no padded / concatenated / transposed copies are made on the way.
    reuse: the PackedMLP this call site built last time (same network, new parameter values): its zero-padded init-table
    buffers are written in place -- two slice copies per table instead of a pad (fill + copy) per block and a concatenation.
    Only for packs that no launch in flight on ANOTHER stream still reads.
    x3 (experiment; chain launches of the 256-wide networks): the weight blocks as two bf16 planes (hip.W_BF16X2) for the split-bf16
    form of the chain engine; the final layer of a pack with n_out <= 32 stays fp32."""
    assert width in (64, 128, 256) and (not x3 or width == 256)
    hid = width // 32
    kin = in_kt_a + in_kt_b
    desc = hip.PsnMlpDesc()
    desc.n_layers = len(layers)
    desc.n_out, desc.out_act, desc.in_kt_a, desc.in_kt_b = n_out, out_act, in_kt_a, in_kt_b
    desc.w_format = hip.W_BF16X2 if x3 else hip.W_F32
    assert len(layers) <= hip.MAX_LAYERS
    plan, biases = [], []
    init_wa, init_wb, init_bias = [], [], []
    direct_init_off = None
    off = b_off = 0
    for li, L in enumerate(layers):
        last = has_final and li == len(layers) - 1
        n_mt = (2 if n_out > 32 else 1) if last else hid  # final layer: 32 outputs, or 64 (256-wide chain engine only)
        rows = n_mt * 32
        lay = desc.layers[li]
        lay.init_off = -1
        bias = L['bias'].detach().float()
        bias_pad = rows - bias.shape[0]  # zero entries behind the real ones (appended in the concatenation below)
        if isinstance(L.get('init_a'), str) and L['init_a'] == DIRECT_INIT:
            assert not last and not init_wa and direct_init_off is None
            lay.init_off = 0
            direct_init_off = 0
        elif L.get('init_a') is not None:
            assert not last
            lay.init_off = width * len(init_wa)
            init_wa.append(L['init_a'])
            if L.get('init_b') is not None:
                init_wb.append(L['init_b'])
            init_bias.append(_pad_vec(bias, rows))  # folded into the init table
            bias, bias_pad = _zeros(rows, device), 0
        n_kt_in = n_kt_act = 0
        tile = n_mt * 1024  # floats per 32-feature k-tile of this layer
        lay.w_off, lay.b_off = off, b_off
        if L.get('w_act') is not None:  # K order of the packed layer: activation tiles first, input-feature tiles last
            # only the k-tiles that hold real columns (7 of 8 behind the 217-output layer of the stage-1 geometry network);
            # the final layer always takes the full width (its stage layout is fixed)
            n_kt_act = hid - 1 if (TRIM_ACT_KTILES and not last and hid > 1 and _src(L['w_act'])[3] <= (hid - 1) * 32) else hid
            plan.append((L['w_act'], n_mt, n_kt_act, off))
            off += n_kt_act * tile
        if L.get('w_in') is not None:
            plan.append((L['w_in'], n_mt, kin, off))
            n_kt_in = kin
            off += kin * tile
        biases.append(bias)
        if bias_pad > 0:
            biases.append(_zeros(bias_pad, device))
        lay.n_kt_in, lay.n_kt_act, lay.n_mt, lay.act = n_kt_in, n_kt_act, n_mt, L['act']
        b_off += rows
    w_buf = torch.empty(max(off, 4), device=device, dtype=torch.float32)
    side = []  # (src, dst) copies of the pack's side tables: ONE launch at the end (hip.copy2d_group) when the call site reuses
    is_zero = [bv.data_ptr() == _zeros(bv.shape[0], device).data_ptr() for bv in biases]  # (cached zeros of one size share one tensor)
    n_bias = sum(bv.shape[0] for bv in biases)
    old_b = getattr(reuse, 'b', None)
    if all(is_zero):
        b_buf = _zeros(n_bias, device)  # all-zero biases of a backward chain
    elif old_b is not None and old_b.numel() == n_bias and old_b.device == w_buf.device and getattr(reuse, 'b_zero', None) == is_zero:
        b_buf, o = old_b, 0  # same layout as last time: the zero segments are in place, the others are rewritten
        for bv, z in zip(biases, is_zero):
            if not z:
                side.append((bv, b_buf[o:o + bv.shape[0]]))
            o += bv.shape[0]
    else:
        b_buf = torch.cat(biases).contiguous()
    group = []
    for w, n_mt, k_tiles, o in plan:
        m, tr, r, c = _src(w)
        assert r <= n_mt * 32 and c <= k_tiles * 32, 'pack_layers: block %dx%d does not fit %dx%d' % (r, c, n_mt * 32, k_tiles * 32)
        group.append((m if m.dtype == torch.float32 else m.float(), tr, n_mt, k_tiles, w_buf[o:o + n_mt * k_tiles * 1024],
                      hip.W_BF16X2 if (x3 and n_mt >= 2) else hip.W_F32))
    hip.mlp_pack_layers(group)  # every block of the network in one launch
    desc.init_stride = width * len(init_wa) if direct_init_off is None else width
    if init_wa:
        def stack(blocks, cols, old):  # [len(blocks) * width, cols], zero padded
            shape = (len(blocks) * width, cols)
            buf = old if (old is not None and tuple(old.shape) == shape and old.device == w_buf.device) else torch.zeros(shape, device=device)
            for i, blk in enumerate(blocks):
                assert blk.shape[0] <= width and blk.shape[1] <= cols
                side.append((blk if blk.dtype == torch.float32 else blk.float(), buf[i * width:i * width + blk.shape[0], :blk.shape[1]]))
            return buf
        n_ib = sum(v.shape[0] for v in init_bias)
        old_ib = getattr(reuse, 'init_bias', None)
        if old_ib is not None and old_ib.numel() == n_ib and old_ib.device == w_buf.device:
            ib, o = old_ib, 0
            for v in init_bias:
                side.append((v, ib[o:o + v.shape[0]]))
                o += v.shape[0]
        else:
            ib = torch.cat(init_bias).contiguous()
        pk = PackedMLP(desc, w_buf, b_buf, stack(init_wa, in_kt_a * 32, getattr(reuse, 'init_wa', None)),
                       stack(init_wb, in_kt_b * 32, getattr(reuse, 'init_wb', None)) if init_wb else None, ib)
    else:
        pk = PackedMLP(desc, w_buf, b_buf)
    pk.b_zero = is_zero
    if side:
        side = [(a_.detach().contiguous() if a_.dim() == 1 else a_.detach(), d_) for a_, d_ in side]
        if all(a_.is_cuda for a_, _ in side):
            hip.copy2d_group(side)  # init-table slices + bias segments: one launch (was a slice copy / a concatenation each)
        else:
            for a_, d_ in side:
                d_.copy_(a_)
    pk.macs_per_row = sum(_src(w)[2] * _src(w)[3] for w, _, _, _ in plan)
    return pk


def pack_relu_mlp(weights, biases, din_a, din_b, skip_at, out_act=hip.OUT_NONE, precompute=True, width=256, reuse=None, x3=False):
    """stage2 Network / Normal_Network (stage2/model/renderer.py:17-49) of width 256: ReLU stack, the
    input is concatenated AFTER layer ``skip_at``.  Input row = [table A (din_a real cols, padded to
    a multiple of 32) | table B (din_b)].  precompute=True evaluates the input block of layer 0 and of the
    skip layer through per-row init tables (see module docstring)."""
    ka = (din_a + 31) // 32
    kb = (din_b + 31) // 32 if din_b > 0 else 0

    def split_in(w):  # [o, din_a + din_b] -> [o, (ka+kb)*32] (single table: the slice itself, the packer zero-fills)
        if kb == 0:
            return w[:, :din_a]
        wa = _pad_cols(w[:, :din_a], ka * 32)
        return torch.cat([wa, _pad_cols(w[:, din_a:din_a + din_b], kb * 32)], dim=1)

    def in_block(W):
        if precompute:
            return dict(init_a=W[:, :din_a], init_b=W[:, din_a:din_a + din_b] if kb else None)
        return dict(w_in=split_in(W))

    layers = []
    n = len(weights)
    for li in range(n):
        W, b = weights[li].detach(), biases[li].detach()
        act = hip.ACT_RELU if li < n - 1 else hip.ACT_NONE
        if li == 0:
            layers.append(dict(w_act=None, bias=b, act=act, **in_block(W)))
        elif li - 1 == skip_at:  # input of this layer is cat[y(256), x]
            layers.append(dict(w_act=W[:, :width], bias=b, act=act, **in_block(W[:, width:])))
        else:
            layers.append(dict(w_act=W, bias=b, act=act))
    assert all(L['bias'].shape[0] == width for L in layers[:-1]), 'fused path: every hidden layer must have the given width'
    return pack_layers(layers, ka, kb, weights[-1].shape[0], out_act, weights[0].device, width=width, reuse=reuse, x3=x3)


class PackedBf16(object):
    """A 256-wide ReLU network packed for the bf16 inference engine (csrc/mlp_infer_bf16.hip)."""

    def __init__(self, desc, w, final_bias):
        self.desc, self.w, self.final_bias = desc, w, final_bias

    def __call__(self, tab_a, n_rows, a_div=1, a_mod=None, tab_b=None, b_div=1, b_mod=1, out=None):
        """tab_a / tab_b: [n, 64] bfloat16 feature tables (the two halves of the input block)."""
        if a_mod is None:
            a_mod = tab_a.shape[0]
        return hip.mlp_infer_bf16(self.desc, self.w, self.final_bias, tab_a, a_div, a_mod, tab_b, b_div, b_mod, n_rows, out=out)


def pack_relu_mlp_bf16(weights, biases, din_a, din_b, skip_at, out_act=hip.OUT_NONE):
    """stage2 Network (stage2/model/renderer.py:34-49) of width 256 for the bf16 engine: the input row is
    [table A (din_a <= 64 real columns) | table B (din_b <= 64)], concatenated again AFTER layer ``skip_at``.
    Weight stream layout: include/psnerf_hip.h (psn_mlp_infer_bf16)."""
    n = len(weights)
    dev = weights[0].device
    assert n - 1 <= hip.MAX_LAYERS and din_a <= 64 and din_b <= 64
    assert all(w.shape[0] == 256 for w in weights[:-1]) and weights[-1].shape[0] <= 32
    desc = hip.PsnBf16Desc()
    desc.n_hidden, desc.n_out, desc.out_act = n - 1, weights[-1].shape[0], out_act
    KS = 8 * 512  # bf16 elements per k-step of a hidden layer
    sizes = []
    for li in range(n - 1):
        has_in = li == 0 or li - 1 == skip_at
        desc.has_in[li] = int(has_in)
        sizes.append(9 * KS if li == 0 else (17 + (8 if has_in else 0)) * KS)
    # the kernel always requests a full 72 KB stage: the final layer's 16 KB block is followed by 56 KB of padding
    buf = torch.zeros(sum(sizes) + 72 * 512, device=dev, dtype=torch.bfloat16)

    def bias_cols(b):
        b = b.detach().float()
        hi = b.to(torch.bfloat16).float()
        return torch.stack([hi, b - hi], dim=1).contiguous()

    def pack_in(W_in, dst):  # [256, din_a + din_b] -> 4 k-steps of table A, 4 of table B
        hip.mlp_pack_bf16(W_in[:, :din_a], False, 8, 0, 4, dst[:4 * KS])
        if din_b > 0:
            hip.mlp_pack_bf16(W_in[:, din_a:din_a + din_b], False, 8, 0, 4, dst[4 * KS:8 * KS])
        else:
            dst[4 * KS:8 * KS].zero_()

    off = 0
    for li in range(n - 1):
        W = weights[li].detach().float()
        dst = buf[off:off + sizes[li]]
        if li == 0:
            pack_in(W, dst)
            hip.mlp_pack_bf16(bias_cols(biases[li]), False, 8, 0, 1, dst[8 * KS:9 * KS])
        else:
            W_act = W[:, :256]
            hip.mlp_pack_bf16(W_act, True, 8, 0, 8, dst[:8 * KS])
            hip.mlp_pack_bf16(bias_cols(biases[li]), False, 8, 0, 1, dst[8 * KS:9 * KS])
            hi = 9 * KS
            if desc.has_in[li]:
                pack_in(W[:, 256:], dst[9 * KS:17 * KS])
                hi = 17 * KS
            hip.mlp_pack_bf16(W_act, True, 8, 8, 8, dst[hi:hi + 8 * KS])
        off += sizes[li]
    hip.mlp_pack_bf16(weights[-1].detach().float(), True, 1, 0, 16, buf[off:off + 16 * 512])
    fb = torch.zeros(32, device=dev)
    fb[:weights[-1].shape[0]] = biases[-1].detach().float()
    return PackedBf16(desc, buf, fb)


class PackedBf16Grouped(object):
    """A 256-wide ReLU network packed for the GROUPED form of the bf16 engine (psn_mlp_infer_bf16_grouped): rows
    (g, n) -> g * Ns + n, the input block of a row is [table A row n | group features of g].  The group's part of every
    input layer is folded into a per-group bias in fp32: V[g, i] = W_b,i @ x_g + b_i (one small product per call)."""

    def __init__(self, desc, w, final_bias, wb_t, b_in):
        self.desc, self.w, self.final_bias = desc, w, final_bias
        self.wb_t = wb_t    # [64, n_in * 256] fp32: the table-B columns of the input layers, transposed, zero rows beyond din_b
        self.b_in = b_in    # [n_in * 256] fp32: the biases of the input layers
        self.n_in = b_in.numel() // 256

    def group_bias(self, tab_b):
        """tab_b [n_groups, 64] fp32 group features -> bias k-steps [n_groups * n_in, 4096] bfloat16."""
        V = hip.gemm(tab_b, self.wb_t, bias=self.b_in, epi=hip.EPI_BIAS)  # [n_groups, n_in * 256]
        return hip.bf16_pack_group_bias(V.view(-1, 256))

    def __call__(self, tab_a, tab_b, out=None):
        """tab_a [Ns, 64] bfloat16, tab_b [n_groups, 64] float32 -> [n_groups * Ns, n_out] (group-major rows)."""
        return hip.mlp_infer_bf16_grouped(self.desc, self.w, self.final_bias, tab_a, self.group_bias(tab_b), tab_b.shape[0], out=out)


def pack_relu_mlp_bf16_grouped(weights, biases, din_a, din_b, skip_at, out_act=hip.OUT_NONE):
    """As pack_relu_mlp_bf16 for the grouped form: the weight stream holds the table-A columns of the input layers only
    (4 k-steps) and no bias k-step for them (include/psnerf_hip.h, psn_mlp_infer_bf16_grouped)."""
    n = len(weights)
    dev = weights[0].device
    assert n - 1 <= hip.MAX_LAYERS and din_a <= 64 and 0 < din_b <= 64
    assert all(w.shape[0] == 256 for w in weights[:-1]) and weights[-1].shape[0] <= 32
    desc = hip.PsnBf16Desc()
    desc.n_hidden, desc.n_out, desc.out_act = n - 1, weights[-1].shape[0], out_act
    KS = 8 * 512
    sizes = []
    for li in range(n - 1):
        has_in = li == 0 or li - 1 == skip_at
        desc.has_in[li] = int(has_in)
        sizes.append(4 * KS if li == 0 else (16 + (4 if has_in else 1)) * KS)
    buf = torch.zeros(sum(sizes) + 72 * 512, device=dev, dtype=torch.bfloat16)

    def bias_cols(b):
        b = b.detach().float()
        hi = b.to(torch.bfloat16).float()
        return torch.stack([hi, b - hi], dim=1).contiguous()

    wb, b_in = [], []
    off = 0
    for li in range(n - 1):
        W = weights[li].detach().float()
        dst = buf[off:off + sizes[li]]
        if li == 0:
            hip.mlp_pack_bf16(W[:, :din_a], False, 8, 0, 4, dst[:4 * KS])
            wb.append(W[:, din_a:din_a + din_b]); b_in.append(biases[li].detach().float())
        else:
            W_act = W[:, :256]
            hip.mlp_pack_bf16(W_act, True, 8, 0, 8, dst[:8 * KS])
            hi = 8 * KS
            if desc.has_in[li]:
                hip.mlp_pack_bf16(W[:, 256:256 + din_a], False, 8, 0, 4, dst[hi:hi + 4 * KS])
                wb.append(W[:, 256 + din_a:256 + din_a + din_b]); b_in.append(biases[li].detach().float())
                hi += 4 * KS
            else:
                hip.mlp_pack_bf16(bias_cols(biases[li]), False, 8, 0, 1, dst[hi:hi + KS])
                hi += KS
            hip.mlp_pack_bf16(W_act, True, 8, 8, 8, dst[hi:hi + 8 * KS])
        off += sizes[li]
    hip.mlp_pack_bf16(weights[-1].detach().float(), True, 1, 0, 16, buf[off:off + 16 * 512])
    fb = torch.zeros(32, device=dev)
    fb[:weights[-1].shape[0]] = biases[-1].detach().float()
    wb_t = torch.zeros(64, len(wb) * 256, device=dev)
    for i, w in enumerate(wb):
        wb_t[:din_b, i * 256:(i + 1) * 256] = w.t()
    return PackedBf16Grouped(desc, buf, fb, wb_t.contiguous(), torch.cat(b_in).contiguous())


def pack_geo_occupancy(weights, biases, skips, d_pe, x3=False):
    """stage1 occupancy-only network (stage1/model/network.py:85-95,124-125): softplus(beta=100)
    stack, before layer l in ``skips`` the input becomes cat[x, pe]/sqrt(2); only output row 0 of the
    last layer is evaluated, followed by sigmoid(-10 x).  Every query point is distinct, so the input
    block stays on the MFMA k-tile path."""
    ka = (d_pe + 31) // 32
    layers = []
    n = len(weights)
    inv = 1.0 / math.sqrt(2.0)
    for li in range(n):
        W, b = weights[li].detach(), biases[li].detach()
        last = li == n - 1
        act = hip.ACT_NONE if last else hip.ACT_SOFTPLUS100
        if last:
            W, b = W[:1], b[:1]
        if li == 0:
            layers.append(dict(w_in=W, w_act=None, bias=b, act=act))
        elif li in skips:
            d_x = W.shape[1] - d_pe
            layers.append(dict(w_in=W[:, d_x:] * inv, w_act=W[:, :d_x] * inv, bias=b, act=act))
        else:
            layers.append(dict(w_in=None, w_act=W, bias=b, act=act))
    return pack_layers(layers, ka, 0, 1, hip.OUT_OCC, weights[0].device, x3=x3)


def pack_relu_bwd(weights, skip_at, width=256, bits=False, x3=False):
    """Backward (d x) chain of a 256-wide ReLU MLP for the fused kernel: chain layer j computes
    d h_{l-1} = W_l[:, :256]^T d z_l for l = n-1-j (transposed weight packs, no bias), followed by the ReLU mask of
    the forward activation h_{l-1} (PSN_ACT_RELU_MASK, masks supplied at call time) and a dump of d z_{l-1}.
    Chain layer 0 has no weights: it starts from the caller's init table d h_{n-2} = g_out W_{n-1}.
    Returns a PackedMLP whose call needs init_a_direct, mask=[h_{n-2}, ..., h_0], save=[dz_{n-2}, ..., dz_0].
    bits: the masks are the sign-bit words of the forward launch (PSN_ACT_RELU_BITS; hip.mlp_infer save_bits) instead of the
    activations themselves -- 32 bytes per row and layer instead of 4 x width."""
    n = len(weights)
    dev = weights[0].device
    zeros = _zeros(width, dev)
    act = hip.ACT_RELU_BITS if bits else hip.ACT_RELU_MASK
    layers = [dict(init_a=DIRECT_INIT, init_b=None, w_act=None, bias=zeros, act=act)]
    for l in range(n - 2, 0, -1):  # forward layers n-2 .. 1 -> their transposed [in(256), out(256)] blocks
        layers.append(dict(w_act=Transposed(weights[l][:, :width]), bias=zeros, act=act))
    return pack_layers(layers, 1, 0, 0, hip.OUT_NONE, dev, has_final=False, width=width, x3=x3 and width == 256)


# --------------------------------------------------------------------------- stage-1 geometry-field chains
def _t(w):
    return Transposed(w)


def pack_geo_chains(weights, biases, skips, d_pe, single_dump=False, x3=False):
    """The four fused chains of ops.GeoFieldFused for the 256-wide softplus geometry network
    (stage1/model/network.py:85-120); ``weights`` are the EFFECTIVE dense matrices with the 1/sqrt(2) of the skip
    layer already folded in.  Returns dict(fwd, sweep, sweep_bwd, value_bwd, value_bwd_nosweep) of PackedMLP.
    ``single_dump`` (experiment, ops.GEO_SINGLE_DUMP): the consumer chains take the dumped softplus outputs A_l where they took the
    dumped sigmoids S_l and re-form s = 1 - exp(-100 a) in their activation programs; the value pass then dumps one tensor per layer.
    ``x3`` (experiment, ops.CHAIN_X3): the four chains' matrix work as three bf16 partial products (pack_layers x3)."""
    mul_aux, mul2, sp_bwd = ((hip.ACT_MUL_AUX_A, hip.ACT_MUL2_A, hip.ACT_SOFTPLUS_BWD_A) if single_dump else
                             (hip.ACT_MUL_AUX, hip.ACT_MUL2, hip.ACT_SOFTPLUS_BWD))
    n = len(weights)
    assert len(skips) == 1 and weights[1].shape[1] == 256 and weights[n - 1].shape[0] == 257
    sk = skips[0]
    dev = weights[0].device
    W = [w.detach() for w in weights]
    b = [x.detach() for x in biases]
    zeros = _zeros(256, dev)
    ka = (d_pe + 31) // 32
    d_a = W[sk].shape[1] - d_pe  # width of the activation part of the skip layer's input (217)

    def fwd_in(l):  # how forward layer l consumes its input
        if l == 0:
            return dict(w_in=W[0], w_act=None)
        if l == sk:
            return dict(w_in=W[l][:, d_a:], w_act=W[l][:, :d_a])
        return dict(w_in=None, w_act=W[l])

    # F1: value pass, dumps a_{l+1} and sigmoid(100 z_l); HEAD = 256 features, final = occupancy logit
    layers = [dict(bias=b[l], act=hip.ACT_SOFTPLUS100, **fwd_in(l)) for l in range(n - 1)]
    layers.append(dict(w_in=None, w_act=W[n - 1][1:], bias=b[n - 1][1:], act=hip.ACT_HEAD))
    layers.append(dict(w_in=None, w_act=W[n - 1][:1], bias=b[n - 1][:1], act=hip.ACT_NONE))
    fwd = pack_layers(layers, ka, 0, 1, hip.OUT_NONE, dev, x3=x3)

    # F2: reverse sweep r_l = (r_{l+1} * s_l) W_l, starting from row 0 of the last layer (init table with one row)
    layers = [dict(init_a=DIRECT_INIT, init_b=None, w_act=None, bias=zeros, act=mul_aux)]
    for l in range(n - 2, 0, -1):
        layers.append(dict(w_act=_t(W[l]), bias=zeros, act=mul_aux))
    # ... and ends with r_0 = u_0 W_0 (256 -> d_pe encoding columns) as a 64-output FINAL layer: 4 output tiles instead of the 16
    # of a hidden-type layer, the [Q, d_pe] result written densely
    layers.append(dict(w_act=_t(W[0]), bias=_zeros(64, dev), act=hip.ACT_NONE))
    assert 32 < d_pe <= 64
    sweep = pack_layers(layers, ka, 0, d_pe, hip.OUT_NONE, dev, x3=x3)

    # B1: adjoint of the sweep: du_l = dR_l W_l^T ; dR_{l+1} = du_l * s_l ; dS_l = du_l * R_{l+1}
    layers = [dict(bias=zeros, act=mul2, **fwd_in(l)) for l in range(n - 1)]
    sweep_bwd = pack_layers(layers, ka, 0, 0, hip.OUT_NONE, dev, has_final=False, x3=x3)

    # B2: adjoint of the value pass: da_l = W_l^T dz_l ; dz_{l-1} = s (da + 100 dS (1 - s))   [or s * da without sweep]
    def value_bwd(act):
        ls = [dict(init_a=DIRECT_INIT, init_b=None, w_act=_t(W[n - 1][1:]), bias=zeros, act=act)]
        for l in range(n - 2, 0, -1):
            ls.append(dict(w_act=_t(W[l]), bias=zeros, act=act))
        return pack_layers(ls, ka, 0, 0, hip.OUT_NONE, dev, has_final=False, x3=x3)

    return dict(fwd=fwd, sweep=sweep, sweep_bwd=sweep_bwd, value_bwd=value_bwd(sp_bwd),
                value_bwd_nosweep=value_bwd(mul_aux), d_a=d_a, single_dump=bool(single_dump))


def pack_app_chains(weights, biases, d_x, x3=False):
    """stage1 appearance network (stage1/model/network.py:98-106, 128-138): ReLU MLP [d_x + 256] -> 256 x (n-1) -> 3
    whose input is cat[x (points, view encoding, normal: d_x <= 64 columns), 256 geometry features].  The features
    enter the first layer as the chain's initial activations (act_init), x as input-feature k-tiles.
    Returns dict(fwd, bwd): fwd dumps the hidden activations, bwd = ReLU backward chain over transposed packs that
    ends with a HEAD layer producing d features and a 3-output final layer producing d normal."""
    n = len(weights)
    dev = weights[0].device
    W = [w.detach() for w in weights]
    b = [x.detach() for x in biases]
    assert W[0].shape[1] == d_x + 256 and all(w.shape == (256, 256) for w in W[1:n - 1]) and d_x <= 64
    ka = (d_x + 31) // 32
    zeros = _zeros(256, dev)
    layers = [dict(w_in=W[0][:, :d_x], w_act=W[0][:, d_x:], bias=b[0], act=hip.ACT_RELU)]
    for l in range(1, n - 1):
        layers.append(dict(w_in=None, w_act=W[l], bias=b[l], act=hip.ACT_RELU))
    layers.append(dict(w_in=None, w_act=W[n - 1], bias=b[n - 1], act=hip.ACT_NONE))
    # (x3: the backward chain only -- it is linear in its operand once the ReLU masks are fixed; a forward pass with 1e-5 errors
    #  flips the units that sit at their kink and the first layers' gradients then differ by 2e-3 of their scale: measured)
    fwd = pack_layers(layers, ka, 0, W[n - 1].shape[0], hip.OUT_NONE, dev)
    ls = [dict(init_a=DIRECT_INIT, init_b=None, w_act=None, bias=zeros, act=hip.ACT_RELU_MASK)]
    for l in range(n - 2, 0, -1):
        ls.append(dict(w_act=_t(W[l]), bias=zeros, act=hip.ACT_RELU_MASK))
    ls.append(dict(w_act=_t(W[0][:, d_x:]), bias=zeros, act=hip.ACT_HEAD))
    # final layer of the backward chain: d normal = d z_0 W_0[:, d_x-3:d_x] (the HEAD layer leaves d z_0 in the
    # activation registers), so the [Q, 256] x [256, 3] product needs no pass over the dumped d z_0
    ls.append(dict(w_act=_t(W[0][:, d_x - 3:d_x]), bias=_zeros(32, dev), act=hip.ACT_NONE))
    bwd = pack_layers(ls, ka, 0, 3, hip.OUT_NONE, dev, x3=x3)
    return dict(fwd=fwd, bwd=bwd)


class PackedX3Grouped(object):
    """A 256-wide ReLU network packed for the split-bf16 ("bf16x6") engine (csrc/mlp_infer_x3.hip; experiment): rows
    (g, n) -> g * Ns + n, input block [table A row n | group features of g].  As in the exact-fp32 engine the layers that read
    the input block start from the fp32 init tables U[n] = W_a x_n and V[g] = W_b x_g + b (two small GEMMs per call)."""

    def __init__(self, desc, w, bias_steps, final_bias, init_wa, init_wb, init_bias, macs_per_row):
        self.desc, self.w, self.bias_steps, self.final_bias = desc, w, bias_steps, final_bias
        self.init_wa, self.init_wb, self.init_bias = init_wa, init_wb, init_bias
        self.macs_per_row = macs_per_row  # the reference network's MACs per row (what an fp32 evaluation multiplies)

    def __call__(self, tab_a, tab_b, out=None):
        """tab_a [Ns, 64] fp32, tab_b [n_groups, 64] fp32 -> [n_groups * Ns, n_out] fp32 (group-major rows)."""
        U = hip.gemm(tab_a, self.init_wa, trans_b=True)
        V = hip.gemm(tab_b, self.init_wb, trans_b=True, bias=self.init_bias, epi=hip.EPI_BIAS)
        return hip.mlp_infer_x3_grouped(self.desc, self.w, self.bias_steps, self.final_bias, U, V, out=out, macs_per_row=self.macs_per_row)


class PackedX3Occ(object):
    """The stage-1 occupancy network packed for the split-bf16 engine (csrc/mlp_infer_x3.hip, OCC variant; opt-in experiment for
    GRADIENT-FREE queries: shadow rays, ray march, root finder inputs).  Same call surface as PackedMLP.on_points."""

    def __init__(self, desc, w, bias_steps, final_bias, skip_layer, pe_first, macs_per_row):
        self.desc, self.w, self.bias_steps, self.final_bias = desc, w, bias_steps, final_bias
        self.skip_layer, self.pe_first, self.macs_per_row = skip_layer, pe_first, macs_per_row

    def march_sweep(self, origin, direction, far, u, omu, near, n_steps, tau, pe_octaves, pe_scale, early_exit=True):
        """The ray-march sweep (hip.march_sweep_x3): points generated and encoded in the kernel, 128-step blocks with early exit."""
        return hip.march_sweep_x3(self.desc, self.w, self.bias_steps, self.final_bias, origin, direction, far, u, omu, near, n_steps, tau,
                                  pe_octaves, pe_scale, self.skip_layer, self.pe_first, early_exit=early_exit, macs_per_row=self.macs_per_row)

    def on_points(self, points, pe_octaves, pe_scale, out=None, n_rows_dev=None, out_rows=None):
        return hip.mlp_infer_x3_occ(self.desc, self.w, self.bias_steps, self.final_bias, points, pe_octaves, pe_scale,
                                    self.skip_layer, self.pe_first, out=out, n_rows_dev=n_rows_dev, out_rows=out_rows,
                                    macs_per_row=self.macs_per_row)


def pack_geo_occupancy_x3(weights, biases, skips, d_pe):
    """stage1 occupancy-only network (stage1/model/network.py:85-95,124-125; same arguments as pack_geo_occupancy) for the
    split-bf16 engine.  Weight stream in execution order: layer 0 = 4 natural-order k-steps of the encoding columns (two 48 KB
    stages), every further hidden layer 16 permuted k-steps, the final layer's output row 0 as one stage; every hidden layer
    has a bias k-step.  The skip layer (input cat[h, pe] / sqrt(2)) is an ordinary 256-input layer: its encoding columns are
    input features pe_first .. pe_first + d_pe - 1, which the kernel fills in (1 / sqrt(2) folded into the weights here)."""
    n = len(weights)
    dev = weights[0].device
    assert len(skips) <= 1 and d_pe <= 48 and 3 <= n <= hip.MAX_LAYERS + 1
    assert all(w.shape[0] <= 256 for w in weights[:-1]) and weights[0].shape[1] == d_pe
    desc = hip.PsnBf16Desc()
    desc.n_hidden, desc.n_out, desc.out_act = n - 1, 1, hip.OUT_OCC
    KS = hip.X3_KS
    final_elems = 16 * 3 * 512
    buf = torch.zeros(4 * KS + (n - 2) * 16 * KS + final_elems + 56 * 512, device=dev, dtype=torch.bfloat16)  # (+ 56 KB: the last request over-reads)
    bias_steps = torch.zeros(n - 1, 4096, device=dev, dtype=torch.bfloat16)
    inv = 1.0 / math.sqrt(2.0)
    off, macs, skip_layer, pe_first = 0, 0, -1, 0
    for li in range(n - 1):
        W, b = weights[li].detach().float(), biases[li].detach().float()
        macs += W.shape[0] * W.shape[1]
        desc.has_in[li] = int(li == 0)
        if li == 0:
            hip.x3_pack(W.contiguous(), False, 8, 0, 4, buf[off:off + 4 * KS])
            off += 4 * KS
        else:
            if li in skips:
                assert W.shape[1] == 256, 'the skip layer reads cat[h, pe]: 256 input features'
                W = W * inv
                skip_layer, pe_first = li, W.shape[1] - d_pe
            hip.x3_pack(W.contiguous(), True, 8, 0, 16, buf[off:off + 16 * KS])
            off += 16 * KS
        bp = torch.zeros(1, 256, device=dev)
        bp[0, :b.numel()] = b
        hip.x3_pack_bias(bp, bias_steps[li:li + 1])
    macs += weights[-1].shape[1]
    hip.x3_pack(weights[-1].detach().float()[:1].contiguous(), True, 1, 0, 16, buf[off:off + final_elems])
    fb = torch.zeros(32, device=dev)
    fb[0] = biases[-1].detach().float()[0]
    return PackedX3Occ(desc, buf, bias_steps, fb, skip_layer, pe_first, macs)


def pack_relu_mlp_x3_grouped(weights, biases, din_a, din_b, skip_at, out_act=hip.OUT_NONE):
    """stage2 Network (stage2/model/renderer.py:34-49) of width 256 for the split-bf16 engine, grouped form.  Weight stream in
    execution order (include/psnerf_hip.h, psn_mlp_infer_x3_grouped)."""
    n = len(weights)
    dev = weights[0].device
    assert 3 <= n <= hip.MAX_LAYERS + 1 and din_a <= 64 and 0 < din_b <= 64
    assert all(w.shape[0] == 256 for w in weights[:-1]) and weights[-1].shape[0] <= 32
    desc = hip.PsnBf16Desc()
    desc.n_hidden, desc.n_out, desc.out_act = n - 1, weights[-1].shape[0], out_act
    KS = hip.X3_KS
    final_elems = 16 * 3 * 512
    buf = torch.zeros((n - 2) * 16 * KS + final_elems + 56 * 512, device=dev, dtype=torch.bfloat16)  # (+ 56 KB: the last request over-reads)
    bias_steps = torch.zeros(n - 1, 4096, device=dev, dtype=torch.bfloat16)
    wa, wb, b_in = [], [], []
    off = 0
    macs = 0
    for li in range(n - 1):
        W = weights[li].detach().float()
        macs += W.shape[0] * W.shape[1]
        has_in = li == 0 or li - 1 == skip_at
        desc.has_in[li] = int(has_in)
        c0 = 0 if li == 0 else 256
        if li > 0:
            hip.x3_pack(W[:, :256], True, 8, 0, 16, buf[off:off + 16 * KS])
            off += 16 * KS
        if has_in:
            wa.append(W[:, c0:c0 + din_a]); wb.append(W[:, c0 + din_a:c0 + din_a + din_b]); b_in.append(biases[li].detach().float())
        else:
            hip.x3_pack_bias(biases[li].detach().float().view(1, 256).contiguous(), bias_steps[li:li + 1])
    macs += weights[-1].shape[0] * weights[-1].shape[1]
    hip.x3_pack(weights[-1].detach().float(), True, 1, 0, 16, buf[off:off + final_elems])
    fb = torch.zeros(32, device=dev)
    fb[:weights[-1].shape[0]] = biases[-1].detach().float()
    init_wa = torch.zeros(len(wa) * 256, 64, device=dev)
    init_wb = torch.zeros(len(wb) * 256, 64, device=dev)
    for i, (a_, b_) in enumerate(zip(wa, wb)):
        init_wa[i * 256:(i + 1) * 256, :din_a] = a_
        init_wb[i * 256:(i + 1) * 256, :din_b] = b_
    return PackedX3Grouped(desc, buf, bias_steps, fb, init_wa, init_wb, torch.cat(b_in).contiguous(), macs)
