"""ctypes binding of libpsnerf_hip.so (C ABI in include/psnerf_hip.h).

Importing this module loads the shared library and FAILS LOUDLY if it is
missing -- there is no CPU or PyTorch fallback on the product path.  Wrappers
take torch tensors only as carriers of device pointers: they validate
device / dtype / contiguity, pass ``data_ptr()`` and the current HIP stream,
and raise RuntimeError with ``psn_last_error()`` on failure.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpsnerf_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'psnerf_amd: %s is missing. Build it with `python -m psnerf_amd.build` (hipcc, gfx950). '
        'There is no fallback path.' % LIB_PATH)
_lib = ctypes.CDLL(LIB_PATH)

c_f = ctypes.c_void_p  # device pointers travel as void*
i64, i32, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_float

MAX_LAYERS = 12
(EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_SOFTPLUS, EPI_MUL_AUX, EPI_MUL_POS, EPI_BIAS_SIGMOID, EPI_ACCUM,
 EPI_MUL2, EPI_SOFTPLUS_BWD, EPI_MUL_AUX_RAW) = range(11)
(ACT_NONE, ACT_RELU, ACT_SOFTPLUS100, ACT_RELU_MASK, ACT_MUL_AUX, ACT_MUL2, ACT_SOFTPLUS_BWD,
 ACT_HEAD, ACT_RELU_BITS, ACT_MUL_AUX_A, ACT_MUL2_A, ACT_SOFTPLUS_BWD_A) = range(12)
OUT_NONE, OUT_SIGMOID, OUT_OCC = range(3)
W_F32, W_BF16X2 = range(2)  # PsnMlpDesc.w_format / PsnPackItem.format


class PsnMlpLayer(ctypes.Structure):
    _fields_ = [('n_kt_in', i32), ('n_kt_act', i32), ('n_mt', i32), ('act', i32), ('w_off', i64), ('b_off', i64),
                ('init_off', i64)]


class PsnMlpDesc(ctypes.Structure):
    _fields_ = [('n_layers', i32), ('n_out', i32), ('out_act', i32), ('in_kt_a', i32), ('in_kt_b', i32),
                ('init_stride', i32), ('w_format', i32), ('layers', PsnMlpLayer * MAX_LAYERS)]


class PsnBf16Desc(ctypes.Structure):
    _fields_ = [('n_hidden', i32), ('n_out', i32), ('out_act', i32), ('reserved', i32),
                ('has_in', ctypes.c_uint8 * (MAX_LAYERS + 4))]


class PsnScatterItem(ctypes.Structure):
    _fields_ = [('rows', ctypes.c_void_p), ('dense', ctypes.c_void_p), ('row_stride', i64), ('col_stride', i64),
                ('B', i32), ('C', i32), ('fill', f32)]


class PsnPackItem(ctypes.Structure):
    _fields_ = [('W', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('ldw', i64), ('rows', i32), ('cols', i32),
                ('transpose', i32), ('n_mt', i32), ('k_tiles', i32), ('format', i32)]


class PsnWnItem(ctypes.Structure):
    _fields_ = [('v', ctypes.c_void_p), ('g', ctypes.c_void_p), ('w', ctypes.c_void_p), ('dw', ctypes.c_void_p),
                ('dv', ctypes.c_void_p), ('dg', ctypes.c_void_p), ('rows', i32), ('cols', i32), ('scale', f32)]


class PsnGemmTnItem(ctypes.Structure):
    _fields_ = [('A', ctypes.c_void_p), ('lda', i64), ('B', ctypes.c_void_p), ('ldb', i64),
                ('A2', ctypes.c_void_p), ('lda2', i64), ('B2', ctypes.c_void_p), ('ldb2', i64),
                ('C', ctypes.c_void_p), ('ldc', i64), ('M', i32), ('N', i32), ('accumulate', i32),
                ('colsum_a', ctypes.c_void_p), ('b_div', i64), ('b_mod', i64),
                ('B_tab2', ctypes.c_void_p), ('ldb_tab2', i64), ('b2_div', i64), ('b2_mod', i64), ('b_split', i32), ('k_rows', i64)]


class PsnPairSumsItem(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('sx', ctypes.c_void_p), ('dWl', ctypes.c_void_p), ('bias', ctypes.c_void_p)]


class PsnCopy2dItem(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('ld_src', i64), ('dst', ctypes.c_void_p), ('ld_dst', i64), ('rows', i32), ('cols', i32)]


COPY2D_MAX = 24


class PsnCopyBytesItem(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n_bytes', i64), ('aligned', i32)]


COPY_BYTES_MAX = 24


class PsnAdamSeg(ctypes.Structure):
    _fields_ = [('offset', i64), ('grad_offset', i64), ('n', i64), ('neg_step_size', f32), ('bias_correction2_sqrt', f32)]


ADAM_MAX_SEGS = 16


class PsnViewBatch(ctypes.Structure):
    _fields_ = [('images', ctypes.c_void_p), ('image_type', i32), ('lut', ctypes.c_void_p),
                ('object_mask', ctypes.c_void_p), ('surface_mask', ctypes.c_void_p),
                ('points', ctypes.c_void_p), ('normal', ctypes.c_void_p), ('visibility', ctypes.c_void_p), ('vis_plus', ctypes.c_void_p),
                ('light_direction', ctypes.c_void_p),
                ('hw', i64), ('width', i32),
                ('lidx', ctypes.c_void_p), ('n_lights', i32),
                ('pix', ctypes.c_void_p), ('pix0', i64), ('n', i64),
                ('vidx', ctypes.c_void_p), ('n_vis', i32),
                ('rgb', ctypes.c_void_p), ('object_mask_out', ctypes.c_void_p), ('surface_mask_out', ctypes.c_void_p), ('uv', ctypes.c_void_p),
                ('points_out', ctypes.c_void_p), ('normal_out', ctypes.c_void_p), ('visibility_out', ctypes.c_void_p),
                ('vis_train_gt', ctypes.c_void_p), ('sampling_idx_out', ctypes.c_void_p), ('light_direction_out', ctypes.c_void_p)]


class PsnRowAdamItem(ctypes.Structure):
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p), ('exp_avg_sq', ctypes.c_void_p),
                ('rows', i64), ('cols', i32), ('one_minus_beta1', f32), ('one_minus_beta2', f32), ('eps', f32), ('step_size', f32)]


MAX_GROUP = 12

# every exported symbol of include/psnerf_hip.h with its signature
SIGNATURES = {
    'psn_last_error': (ctypes.c_char_p, []),
    'psn_version': (i32, []),
    'psn_composite_fwd': (i32, [c_f, c_f, i64, i32, i32, c_f, c_f, c_f, c_f]),
    'psn_composite_bwd': (i32, [c_f, c_f, c_f, c_f, i64, i32, i32, c_f, c_f, c_f]),
    'psn_pe_encode': (i32, [c_f, i64, i32, f32, c_f, i32, c_f]),
    'psn_pe_encode_bwd': (i32, [c_f, c_f, i64, i32, f32, i32, c_f, i32, c_f, c_f]),
    'psn_app_input': (i32, [c_f, c_f, c_f, i64, i32, c_f, c_f]),
    'psn_pe_encode_jvp': (i32, [c_f, c_f, i64, i32, f32, c_f, i32, c_f]),
    'psn_gemm': (i32, [i32, i32, i64, i32, i32, c_f, i64, c_f, i64, c_f, i64, c_f, i32, c_f, i64, c_f, i64, c_f, i64,
                       i32, c_f, c_f, c_f]),
    'psn_gemm_tn_grouped': (i32, [i32, ctypes.c_void_p, i64, i32, c_f, i64, c_f]),
    'psn_gemm_tn_x3_set_products': (i32, [i32]),
    'psn_mlp_block_order': (i32, [i32]),
    'psn_gemm_tn_grouped_x3': (i32, [i32, ctypes.c_void_p, i64, i32, c_f, i64, c_f]),
    'psn_colsum': (i32, [c_f, c_f, i32, i64, i64, i32, i64, c_f, i32, c_f, c_f]),
    'psn_sample_points': (i32, [c_f, c_f, c_f, c_f, c_f, i64, i32, f32, f32, c_f, c_f, i32, c_f, c_f, i32, c_f, c_f, c_f]),
    'psn_sample_points_flagged': (i32, [c_f, c_f, c_f, c_f, c_f, i64, f32, f32, c_f, c_f, i32, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f]),
    'psn_mlp_pack_layer': (i32, [c_f, i64, i32, i32, i32, i32, i32, c_f, c_f]),
    'psn_mlp_pack_layers': (i32, [i32, ctypes.c_void_p, c_f]),
    'psn_sg_shade_fwd': (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, f32, c_f, i32, i64, i32, i32, c_f, c_f, c_f]),
    'psn_sg_shade_bwd': (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, f32, c_f, i32, i64, i32, i32, c_f, c_f, c_f, c_f,
                               c_f, c_f, c_f, c_f, c_f, c_f]),
    'psn_mf_shade_fwd': (i32, [c_f, c_f, c_f, c_f, c_f, c_f, f32, f32, c_f, i32, i64, c_f, c_f]),
    'psn_mf_shade_bwd': (i32, [c_f, c_f, c_f, c_f, c_f, c_f, f32, f32, c_f, i32, i64, c_f, c_f, c_f, c_f, c_f, c_f,
                               c_f, c_f, c_f]),
    'psn_mlp_infer': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, i64, i64, c_f, i64, i64, c_f, c_f,
                            ctypes.POINTER(ctypes.c_void_p), i64, ctypes.POINTER(ctypes.c_void_p),
                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_f, i64, c_f, c_f, i32,
                            ctypes.POINTER(ctypes.c_uint32), i64, c_f, c_f]),
    'psn_mlp_infer_padded': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, i64, i64, c_f, i64, i64, c_f, c_f,
                                   ctypes.POINTER(ctypes.c_void_p), i64, i64, c_f, c_f, i64, c_f]),
    'psn_mlp_infer_bits': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, i64, i64, c_f, i64, i64, c_f, c_f,
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), i64, i64, c_f, c_f, i64, c_f]),
    'psn_scatter_rows': (i32, [i32, ctypes.c_void_p, c_f, i64, i64, c_f]),
    'psn_gather_rows': (i32, [i32, ctypes.c_void_p, c_f, i64, i64, c_f]),
    'psn_gather_rows_valid': (i32, [i32, ctypes.c_void_p, c_f, c_f, i64, i64, c_f]),
    'psn_surface_index': (i32, [c_f, i64, i64, c_f, c_f, c_f]),
    'psn_view_batch': (i32, [ctypes.c_void_p, c_f]),
    'psn_secant_step': (i32, [c_f, f32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, c_f]),
    'psn_first_crossing': (i32, [c_f, c_f, c_f, c_f, f32, f32, i64, i32, c_f, c_f, c_f]),
    'psn_stage2_loss_fwd': (i32, [c_f, c_f, i32, c_f, c_f, c_f, c_f, i32, c_f, c_f, i32, c_f, c_f, c_f, c_f, c_f, i64, i32, c_f, c_f, c_f,
                                  c_f, c_f, c_f]),
    'psn_stage2_loss_bwd': (i32, [c_f, c_f, c_f, i32, f32, c_f, c_f, c_f, f32, c_f, c_f, c_f, c_f, i32, f32, c_f, c_f, c_f, c_f, i32, f32,
                                  c_f, c_f, c_f, c_f, f32, f32, c_f, c_f, c_f, c_f, i64, i32, c_f, c_f]),
    'psn_pair_sums': (i32, [c_f, i32, i64, i32, c_f, c_f, ctypes.POINTER(ctypes.c_int), c_f]),
    'psn_pair_sums_group_workspace': (i64, [i32, i32, i64, i32]),
    'psn_pair_sums_group': (i32, [i32, ctypes.c_void_p, i32, i64, i32, c_f, i64, i32, i64, c_f, c_f]),
    'psn_row_adam': (i32, [i32, ctypes.c_void_p, c_f, i32, c_f]),
    'psn_row_adam_dev': (i32, [i32, ctypes.c_void_p, c_f, i32, c_f, c_f]),
    'psn_shadow_points': (i32, [c_f, c_f, i64, i32, i32, f32, f32, c_f, c_f, f32, c_f, c_f, c_f, c_f]),
    'psn_mlp_infer_pe': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, i64, i32, f32, c_f, c_f]),
    'psn_mlp_infer_pe_indirect': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, i64, c_f, c_f, i32, f32, c_f, c_f]),
    'psn_root_find': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, c_f, c_f, i64, f32, i32, i32, f32, c_f, c_f]),
    'psn_march_sweep': (i32, [ctypes.POINTER(PsnMlpDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, f32, i64, i32, f32, i32, f32, c_f, c_f, c_f, c_f]),
    'psn_normalize_rows_fwd': (i32, [c_f, i64, f32, c_f, c_f]),
    'psn_normalize_rows_bwd': (i32, [c_f, c_f, i64, f32, c_f, c_f]),
    'psn_light_rows_fwd': (i32, [c_f, c_f, c_f, i32, f32, c_f, c_f, c_f]),
    'psn_light_rows_bwd': (i32, [c_f, c_f, i32, i64, f32, c_f, c_f, c_f, c_f, c_f]),
    'psn_camera_rays': (i32, [c_f, c_f, c_f, c_f, i64, f32, c_f, c_f]),
    'psn_stage1_loss_partial_floats': (i32, []),
    'psn_stage1_loss_fwd': (i32, [c_f] * 10 + [i64, i64, ctypes.c_void_p, c_f, c_f, c_f, c_f]),
    'psn_stage1_loss_terms': (i32, [c_f, i64, ctypes.c_void_p, i32, i32, i32, c_f, c_f]),
    'psn_stage1_loss_bwd': (i32, [c_f] * 11 + [i64, i64, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_f]),
    'psn_surface_normals_fwd': (i32, [c_f, c_f, i64, f32, c_f, c_f, c_f]),
    'psn_surface_normals_bwd': (i32, [c_f, c_f, i64, f32, c_f, c_f, c_f, c_f]),
    'psn_stage1_rays': (i32, [c_f, c_f, i32, c_f, f32, i64, c_f, c_f, c_f, c_f]),
    'psn_surface_points': (i32, [c_f, c_f, c_f, c_f, i64, c_f, c_f, c_f, c_f, c_f]),
    'psn_stage1_targets': (i32, [c_f, i64, i32, i32, c_f, c_f, c_f, c_f, c_f, c_f, i32, f32, c_f, c_f, c_f, c_f, c_f, c_f]),
    'psn_copy2d_group': (i32, [i32, ctypes.c_void_p, c_f]),
    'psn_copy_bytes_group': (i32, [i32, ctypes.c_void_p, c_f]),
    'psn_mask_count': (i32, [c_f, c_f, i64, c_f, c_f]),
    'psn_inverse_index': (i32, [c_f, i64, i64, c_f, c_f, c_f]),
    'psn_adam_flat': (i32, [c_f, c_f, c_f, c_f, i32, ctypes.c_void_p, f32, f32, f32, f32, c_f]),
    'psn_adam_flat_dev': (i32, [c_f, c_f, c_f, c_f, i32, ctypes.c_void_p, f32, f32, f32, f32, c_f, c_f]),
    'psn_weight_norm_fwd': (i32, [i32, ctypes.c_void_p, c_f]),
    'psn_weight_norm_bwd': (i32, [i32, ctypes.c_void_p, c_f]),
    'psn_mlp_pack_bf16': (i32, [c_f, i64, i32, i32, i32, i32, i32, i32, c_f, c_f]),
    'psn_mlp_infer_bf16': (i32, [ctypes.POINTER(PsnBf16Desc), c_f, c_f, c_f, i64, i64, c_f, i64, i64, i64, c_f, c_f]),
    'psn_mlp_infer_bf16_grouped': (i32, [ctypes.POINTER(PsnBf16Desc), c_f, c_f, c_f, i64, c_f, i64, c_f, c_f]),
    'psn_bf16_pack_group_bias': (i32, [c_f, i64, c_f, c_f]),
    'psn_x3_pack': (i32, [c_f, i64, i32, i32, i32, i32, i32, i32, c_f, c_f]),
    'psn_x3_pack_bias': (i32, [c_f, i64, c_f, c_f]),
    'psn_mlp_infer_x3_grouped': (i32, [ctypes.POINTER(PsnBf16Desc), c_f, c_f, c_f, c_f, i64, c_f, i64, c_f, c_f]),
    'psn_mlp_infer_x3_occ': (i32, [ctypes.POINTER(PsnBf16Desc), c_f, c_f, c_f, c_f, i64, c_f, c_f, i32, f32, i32, i32, c_f, c_f]),
    'psn_march_sweep_x3': (i32, [ctypes.POINTER(PsnBf16Desc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, f32, i64, i32, f32, i32, f32, i32, i32,
                                 c_f, c_f, c_f, c_f]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = library out of date: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


# When set to a list (bench.py), kernel wrappers that support it append
# (name, units, start_event, end_event, flops or None) recorded on the launch stream; units = rows for the MLP engines,
# algorithmic FLOPs for the grouped weight-gradient GEMM, algorithmic bytes for the composite kernels.
PROFILE_EVENTS = None


class _Prof(object):
    """``with _Prof(name, units):`` brackets one C-ABI launch with HIP events on the launch stream when PROFILE_EVENTS is
    a list (bench.py); ``units`` = rows, or algorithmic FLOPs / bytes of the launch, as the name's consumer defines."""

    def __init__(self, name, units, flops=None):
        self.name, self.units, self.flops, self.prof = name, units, flops, PROFILE_EVENTS

    def __enter__(self):
        if self.prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.prof is not None and exc[0] is None:
            self.e1.record()
            self.prof.append((self.name, self.units, self.e0, self.e1, self.flops))
        return False


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, _lib.psn_last_error().decode()))


def _ptr(t, name, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise RuntimeError('%s: tensor required' % name)
    if not t.is_cuda:
        raise RuntimeError('%s: must be a HIP device tensor (the product path has no CPU fallback)' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('%s: must be float32, got %s' % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('%s: must be contiguous' % name)
    return t.data_ptr()


def _bits_ptr(t, name):
    """Device pointer of a tensor of sign-bit words ([rows, 4] int64, contiguous)."""
    if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == 4):
        raise RuntimeError('%s: sign-bit words are a contiguous int64 [rows, 4] device tensor' % name)
    return t.data_ptr()


def _iptr(t, name, allow_none=False):
    """Device pointer of an int64 index tensor."""
    if t is None:
        if allow_none:
            return None
        raise RuntimeError('%s: tensor required' % name)
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
        raise RuntimeError('%s: must be a contiguous int64 HIP device tensor' % name)
    return t.data_ptr()


def _stream():
    # the raw handle of the current stream of the current device: torch.cuda.current_stream().cuda_stream builds a Stream object
    # through three Python layers (11 us; a step asks 30 - 65 times), the two C calls below take ~1 us
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def version():
    return _lib.psn_version()


# --------------------------------------------------------------------------- composite
def composite_fwd(alpha, rgb, white_bg, need_weights=True):
    """alpha [N,S], rgb [N,S,3] or None -> (weights [N,S] or None, rgb_out [N,3] or None, acc [N])."""
    N, S = alpha.shape
    weights = torch.empty_like(alpha) if need_weights else None
    rgb_out = torch.empty(N, 3, device=alpha.device, dtype=torch.float32) if rgb is not None else None
    acc = torch.empty(N, device=alpha.device, dtype=torch.float32)
    # algorithmic bytes: alpha 4 S (+ colour 12 S) in, weights 4 S (if kept) + rgb 12 + acc 4 out
    nbytes = N * (4 * S + (12 * S + 12 if rgb is not None else 0) + (4 * S if need_weights else 0) + 4)
    with _Prof('composite_fwd' if rgb is not None else 'composite_alpha', nbytes):
        _check(_lib.psn_composite_fwd(_ptr(alpha, 'alpha'), _ptr(rgb, 'rgb', True), N, S, int(bool(white_bg)),
                                      _ptr(weights, 'weights', True), _ptr(rgb_out, 'rgb_out', True), _ptr(acc, 'acc'),
                                      _stream()), 'composite_fwd')
    return weights, rgb_out, acc


def composite_bwd(alpha, rgb, d_rgb_out, d_acc, white_bg):
    N, S = alpha.shape
    d_alpha = torch.empty_like(alpha)
    d_rgb = torch.empty_like(rgb) if rgb is not None else None
    with _Prof('composite_bwd', N * (36 * S + 16)):  # 36 S + 16 bytes per ray
        _check(_lib.psn_composite_bwd(_ptr(alpha, 'alpha'), _ptr(rgb, 'rgb', True), _ptr(d_rgb_out, 'd_rgb_out', True),
                                      _ptr(d_acc, 'd_acc', True), N, S, int(bool(white_bg)), _ptr(d_alpha, 'd_alpha'),
                                      _ptr(d_rgb, 'd_rgb', True), _stream()), 'composite_bwd')
    return d_alpha, d_rgb


# --------------------------------------------------------------------------- positional encoding
def pe_encode(x, n_freqs, out_stride=None, scale=1.0):
    """x [n,3] -> [n, out_stride] = [x, sin(2^k x), cos(2^k x)]_k, zero padded to out_stride."""
    n = x.shape[0]
    width = 3 + 6 * n_freqs
    out_stride = width if out_stride is None else out_stride
    out = torch.empty(n, out_stride, device=x.device, dtype=torch.float32)
    _check(_lib.psn_pe_encode(_ptr(x, 'x'), n, n_freqs, float(scale), _ptr(out, 'out'), out_stride, _stream()),
           'pe_encode')
    return out


def app_input(p, v, normal, n_freqs):
    """[Q, 64] input table of the stage-1 appearance chain: [p | gamma(v / |v|) | normal | 0] (psn_app_input)."""
    Q = p.shape[0]
    assert p.shape == v.shape == normal.shape == (Q, 3)
    out = torch.empty(Q, 64, device=p.device, dtype=torch.float32)
    p, v, normal = p.contiguous(), v.contiguous(), normal.contiguous()
    if Q:
        _check(_lib.psn_app_input(_ptr(p, 'p'), _ptr(v, 'v'), _ptr(normal, 'normal'), Q, int(n_freqs), out.data_ptr(), _stream()),
               'app_input')
    return out


def pe_encode_jvp(x, t, n_freqs, out_stride, scale=1.0):
    """J_PE(x) t: tangent t [n,3] -> [n, out_stride]."""
    n = x.shape[0]
    out = torch.empty(n, out_stride, device=x.device, dtype=torch.float32)
    _check(_lib.psn_pe_encode_jvp(_ptr(x, 'x'), _ptr(t, 't'), n, n_freqs, float(scale), _ptr(out, 'out'), out_stride,
                                  _stream()), 'pe_encode_jvp')
    return out


def pe_encode_bwd(x, d_out, n_freqs, scale=1.0, add=None):
    """d_out (and ``add``, a second gradient summed in first): row-major views [n, >= 3 + 6 n_freqs] -- column ranges of
    wider tensors are fine (row stride = stride(0))."""
    n = x.shape[0]
    d_x = torch.empty_like(x)
    assert d_out.shape[1] >= 3 + 6 * n_freqs and (add is None or add.shape[1] >= 3 + 6 * n_freqs)
    _check(_lib.psn_pe_encode_bwd(_ptr(x, 'x'), _mat_ptr(d_out, 'd_out'), n, n_freqs, float(scale), _ld(d_out),
                                  None if add is None else _mat_ptr(add, 'add'), 0 if add is None else _ld(add),
                                  _ptr(d_x, 'd_x'), _stream()), 'pe_encode_bwd')
    return d_x


def sample_points(origin, direction, far, out, hit, near, u0, idx=None, dist=None, delta=0.0, u1=None, noise=None):
    """Fill rows ``idx`` (all rows if None) of out [N,S,3] with origin + direction * depth profile (psn_sample_points).
    u0 / u1: (linspace(0,1,c), 1 - linspace) pairs of device tensors."""
    n = origin.shape[0] if idx is None else idx.shape[0]
    c0 = u0[0].numel()
    c1 = 0 if u1 is None else u1[0].numel()
    assert out.shape[1] == c0 + c1 and out.is_contiguous() and (idx is None or idx.dtype == torch.int64)
    if noise is not None:
        assert noise.numel() == n * (c0 + c1)
    _check(_lib.psn_sample_points(_ptr(origin, 'origin'), _ptr(direction, 'direction'), _ptr(dist, 'dist', True), _ptr(far, 'far'),
                                  None if idx is None else idx.data_ptr(), n, int(bool(hit)), float(near), float(delta),
                                  _ptr(u0[0], 'u0'), _ptr(u0[1], 'omu0'), c0,
                                  None if u1 is None else _ptr(u1[0], 'u1'), None if u1 is None else _ptr(u1[1], 'omu1'), c1,
                                  _ptr(noise, 'noise', True), _ptr(out, 'out'), _stream()), 'sample_points')
    return out


def sample_points_flagged(origin, direction, dist, far, flags, out, near, delta, u0, u1, u_miss, noise=None):
    """Every ray of out [N,S,3] in one launch: hit profile where flags (bool [N]) is set, free-space profile elsewhere
    (psn_sample_points_flagged).  u0 / u1 / u_miss: (linspace, 1 - linspace) pairs; u1 may be None (S = c0)."""
    n = origin.shape[0]
    c0 = u0[0].numel()
    c1 = 0 if u1 is None else u1[0].numel()
    assert out.shape[1] == c0 + c1 == u_miss[0].numel() and out.is_contiguous()
    assert flags.dtype == torch.bool and flags.is_cuda and flags.is_contiguous() and flags.numel() == n
    if noise is not None:
        assert noise.numel() == n * (c0 + c1)
    _check(_lib.psn_sample_points_flagged(_ptr(origin, 'origin'), _ptr(direction, 'direction'), _ptr(dist, 'dist'), _ptr(far, 'far'),
                                          flags.data_ptr(), n, float(near), float(delta), _ptr(u0[0], 'u0'), _ptr(u0[1], 'omu0'), c0,
                                          None if u1 is None else _ptr(u1[0], 'u1'), None if u1 is None else _ptr(u1[1], 'omu1'), c1,
                                          _ptr(u_miss[0], 'um'), _ptr(u_miss[1], 'omum'), _ptr(noise, 'noise', True), _ptr(out, 'out'),
                                          _stream()), 'sample_points_flagged')
    return out


# --------------------------------------------------------------------------- GEMM
_ws_cache = {}


SCATTER_MAX_ITEMS = 16


def copy2d_group(pairs):
    """pairs: [(src, dst)] fp32 device tensors of equal shape, 1-D (contiguous) or 2-D with unit column stride (row strides
    allowed: column slices of a parameter, row blocks of a table) -> dst[...] = src[...], COPY2D_MAX pairs per launch."""
    for c0 in range(0, len(pairs), COPY2D_MAX):
        chunk = pairs[c0:c0 + COPY2D_MAX]
        arr = (PsnCopy2dItem * len(chunk))()
        for e, (src, dst) in zip(arr, chunk):
            assert src.shape == dst.shape and src.dtype == dst.dtype == torch.float32 and src.is_cuda and dst.is_cuda and src.numel() > 0
            if src.dim() == 1:
                assert src.stride(0) == 1 and dst.stride(0) == 1
                e.rows, e.cols, e.ld_src, e.ld_dst = 1, src.shape[0], src.shape[0], src.shape[0]
            else:
                assert src.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
                e.rows, e.cols, e.ld_src, e.ld_dst = src.shape[0], src.shape[1], src.stride(0), dst.stride(0)
            e.src, e.dst = src.data_ptr(), dst.data_ptr()
        _check(_lib.psn_copy2d_group(len(chunk), ctypes.addressof(arr), _stream()), 'copy2d_group')


def copy_group(pairs):
    """dst.copy_(src) for every (dst, src) pair of contiguous device tensors with equal dtype and element count, <= 24 per launch
    (psn_copy_bytes_group): the batch of a step into the input buffers of a captured HIP graph in one launch."""
    for c0 in range(0, len(pairs), COPY_BYTES_MAX):
        chunk = pairs[c0:c0 + COPY_BYTES_MAX]
        arr = (PsnCopyBytesItem * len(chunk))()
        for e, (dst, src) in zip(arr, chunk):
            assert dst.is_cuda and src.is_cuda and dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
            e.src, e.dst, e.n_bytes = src.data_ptr(), dst.data_ptr(), dst.numel() * dst.element_size()
        _check(_lib.psn_copy_bytes_group(len(chunk), ctypes.addressof(arr), _stream()), 'copy_bytes_group')


def mask_count(mask_a, mask_b=None, out=None):
    """Number of elements with mask_a & mask_b (torch.bool tensors of one size) -> float32 device tensor [1], one launch."""
    assert mask_a.is_cuda and mask_a.dtype == torch.bool and mask_a.is_contiguous()
    if mask_b is not None:
        assert mask_b.is_cuda and mask_b.dtype == torch.bool and mask_b.is_contiguous() and mask_b.numel() == mask_a.numel()
    if out is None:
        out = torch.empty(1, device=mask_a.device, dtype=torch.float32)
    assert out.is_cuda and out.dtype == torch.float32 and out.numel() == 1
    _check(_lib.psn_mask_count(mask_a.data_ptr(), None if mask_b is None else mask_b.data_ptr(), mask_a.numel(), out.data_ptr(), _stream()),
           'mask_count')
    return out


def inverse_index(idx, n_pixels, count=None):
    """Pixel -> row map of an ascending index list: [n_pixels] int32, -1 where the pixel is not in idx (one launch).  count
    (float32 [1] on the device): the list is padded to a fixed length and only its first count[0] entries are real."""
    assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 1
    assert count is None or (count.is_cuda and count.dtype == torch.float32 and count.numel() == 1)
    inv = torch.empty(n_pixels, device=idx.device, dtype=torch.int32)
    _check(_lib.psn_inverse_index(idx.data_ptr(), idx.numel(), n_pixels, inv.data_ptr(), None if count is None else count.data_ptr(), _stream()),
           'inverse_index')
    return inv


def scatter_rows(specs, rows, inv, n_pixels, n_surf):
    """specs: [(B, C, fill)], rows: matching list of [B*Ns, C] fp32 tensors (any strides, stride-0 columns allowed) ->
    list of dense [B, N, C] tensors, one launch per 16 outputs."""
    dev = inv.device
    dense = [torch.empty(B, n_pixels, C, device=dev, dtype=torch.float32) for B, C, _ in specs]
    for c0 in range(0, len(specs), SCATTER_MAX_ITEMS):
        n = min(SCATTER_MAX_ITEMS, len(specs) - c0)
        arr = (PsnScatterItem * n)()
        for i in range(n):
            (B, C, fill), r, e = specs[c0 + i], rows[c0 + i], arr[i]
            assert r.dtype == torch.float32 and r.is_cuda and r.shape == (B * n_surf, C)
            e.rows, e.dense = r.data_ptr(), dense[c0 + i].data_ptr()
            e.row_stride, e.col_stride, e.B, e.C, e.fill = r.stride(0), r.stride(1), B, C, float(fill)
        assert inv.dtype == torch.int32 and inv.is_contiguous() and inv.numel() == n_pixels
        _check(_lib.psn_scatter_rows(n, ctypes.addressof(arr), inv.data_ptr(), n_pixels, n_surf, _stream()), 'scatter_rows')
    return dense


def surface_index(mask, capacity):
    """Ascending positions of the True elements of a 1-D bool device tensor in a FIXED-size int64 list [capacity]: entries behind
    the real ones repeat the last one (psn_surface_index; no host synchronisation).  -> (idx, count float32 [1])."""
    assert mask.is_cuda and mask.dtype == torch.bool and mask.dim() == 1 and mask.is_contiguous()
    idx = torch.empty(int(capacity), device=mask.device, dtype=torch.int64)
    count = torch.empty(1, device=mask.device, dtype=torch.float32)
    _check(_lib.psn_surface_index(mask.data_ptr(), mask.numel(), int(capacity), idx.data_ptr(), count.data_ptr(), _stream()), 'surface_index')
    return idx, count


IMAGE_TYPES = {torch.float32: 0, torch.uint8: 1, torch.uint16: 2}


def view_batch(tables, lidx, pix, n, out, pix0=0, vidx=None):
    """One stage-2 training batch gathered from the resident tables of a view (psn_view_batch; one launch, no allocation).
    ``tables``: dict of contiguous device tensors -- 'images' [Lv, hw, 3] (float32 / uint8 / uint16), 'lut' (integer images),
    'object_mask' / 'surface_mask' [hw] bool, 'points' / 'normal' [hw, 3], optional 'visibility' [Lv, hw], 'vis_plus' [R, hw],
    'light_direction' [Lv, 3]; 'width'.  ``lidx`` [L] / ``pix`` [n] (or None: the range pix0 .. pix0 + n) / ``vidx`` [V]: device int64.
    ``out``: dict of preallocated contiguous outputs, any subset of 'rgb' [L, n, 3], 'object_mask' / 'surface_mask' [n] bool,
    'uv' [n, 2], 'points' / 'normal' [n, 3], 'visibility' [L, n], 'vis_train_gt' [V, n], 'sampling_idx' [n] int64,
    'light_direction' [L, 3]."""
    b = PsnViewBatch()
    img = tables['images']
    assert img.is_cuda and img.is_contiguous() and img.dtype in IMAGE_TYPES and img.dim() == 3 and img.shape[2] == 3
    hw = img.shape[1]
    b.images, b.image_type, b.hw, b.width = img.data_ptr(), IMAGE_TYPES[img.dtype], hw, int(tables['width'])
    if b.image_type:
        lut = tables['lut']
        assert lut.is_cuda and lut.dtype == torch.float32 and lut.is_contiguous() and lut.numel() == (256 if b.image_type == 1 else 65536)
        b.lut = lut.data_ptr()

    def table(key, shape_tail, dtype):
        t = tables.get(key)
        if t is None:
            return None
        assert t.is_cuda and t.is_contiguous() and t.dtype == dtype and tuple(t.shape[-len(shape_tail):]) == shape_tail, key
        return t.data_ptr()
    b.object_mask, b.surface_mask = table('object_mask', (hw,), torch.bool), table('surface_mask', (hw,), torch.bool)
    b.points, b.normal = table('points', (hw, 3), torch.float32), table('normal', (hw, 3), torch.float32)
    b.visibility, b.vis_plus = table('visibility', (hw,), torch.float32), table('vis_plus', (hw,), torch.float32)
    b.light_direction = table('light_direction', (3,), torch.float32)
    L = 0 if lidx is None else int(lidx.numel())
    for t_ in (lidx, pix, vidx):
        assert t_ is None or (t_.is_cuda and t_.dtype == torch.int64 and t_.is_contiguous() and t_.dim() == 1)
    assert pix is None or pix.numel() == n
    b.lidx, b.n_lights = (None if lidx is None else lidx.data_ptr()), L
    b.pix, b.pix0, b.n = (None if pix is None else pix.data_ptr()), int(pix0), int(n)
    V = 0 if (vidx is None or 'vis_train_gt' not in out) else int(vidx.numel())
    b.vidx, b.n_vis = (vidx.data_ptr() if V else None), V
    want = {'rgb': ((L, n, 3), torch.float32), 'object_mask': ((n,), torch.bool), 'surface_mask': ((n,), torch.bool),
            'uv': ((n, 2), torch.float32), 'points': ((n, 3), torch.float32), 'normal': ((n, 3), torch.float32),
            'visibility': ((L, n), torch.float32), 'vis_train_gt': ((V, n), torch.float32), 'sampling_idx': ((n,), torch.int64),
            'light_direction': ((L, 3), torch.float32)}
    field = {'object_mask': 'object_mask_out', 'surface_mask': 'surface_mask_out', 'points': 'points_out', 'normal': 'normal_out',
             'visibility': 'visibility_out', 'sampling_idx': 'sampling_idx_out', 'light_direction': 'light_direction_out'}
    for k, t in out.items():
        shape, dt = want[k]
        assert t.is_cuda and t.is_contiguous() and t.dtype == dt and t.numel() == int(torch.Size(shape).numel()), (k, tuple(t.shape), shape)
        setattr(b, field.get(k, k), t.data_ptr())
    _check(_lib.psn_view_batch(ctypes.addressof(b), _stream()), 'view_batch')
    return out


def gather_rows(specs, dense_grads, idx, n_pixels, n_surf, inv=None):
    """Adjoint of scatter_rows for the given dense gradients (contiguous [B, N, C]) -> list of [B*Ns, C] tensors.  inv (the
    pixel -> row map of the scatter): rows of a padded index list that are not the first of their pixel receive zeros."""
    out = [torch.empty(B * n_surf, C, device=idx.device, dtype=torch.float32) for B, C, _ in specs]
    for c0 in range(0, len(specs), SCATTER_MAX_ITEMS):
        n = min(SCATTER_MAX_ITEMS, len(specs) - c0)
        arr = (PsnScatterItem * n)()
        for i in range(n):
            (B, C, _), g, e = specs[c0 + i], dense_grads[c0 + i], arr[i]
            assert g.is_contiguous() and g.shape == (B, n_pixels, C)
            e.rows, e.dense, e.B, e.C = out[c0 + i].data_ptr(), _ptr(g, 'dense_grad'), B, C
        assert idx.dtype == torch.int64 and idx.is_contiguous()
        if inv is not None:
            assert inv.dtype == torch.int32 and inv.is_contiguous() and inv.numel() == n_pixels
            _check(_lib.psn_gather_rows_valid(n, ctypes.addressof(arr), idx.data_ptr(), inv.data_ptr(), n_pixels, n_surf, _stream()),
                   'gather_rows_valid')
            continue
        _check(_lib.psn_gather_rows(n, ctypes.addressof(arr), idx.data_ptr(), n_pixels, n_surf, _stream()), 'gather_rows')
    return out


def secant_step(occ, tau, d_pred, d_low, d_high, f_low, f_high, origin, direction, p_mid):
    """One regula-falsi iteration in place (csrc/sample.hip); occ=None: initial step."""
    _check(_lib.psn_secant_step(_ptr(occ, 'occ', True), float(tau), _ptr(d_pred, 'd_pred'), _ptr(d_low, 'd_low'),
                                _ptr(d_high, 'd_high'), _ptr(f_low, 'f_low'), _ptr(f_high, 'f_high'),
                                _ptr(origin, 'origin', True), _ptr(direction, 'direction', True), _ptr(p_mid, 'p_mid', True),
                                d_pred.numel(), _stream()), 'secant_step')


def first_crossing(occ, far, u, omu, near, tau):
    """occ [N, M] sweep occupancies -> (bracket [4, N] float32, flags [N] int32) -- see psn_first_crossing."""
    N, M = occ.shape
    bracket = torch.empty(4, N, device=occ.device, dtype=torch.float32)
    flags = torch.empty(N, device=occ.device, dtype=torch.int32)
    _check(_lib.psn_first_crossing(_ptr(occ, 'occ'), _ptr(far, 'far'), _ptr(u, 'u'), _ptr(omu, 'omu'), float(near), float(tau),
                                   N, M, bracket.data_ptr(), flags.data_ptr(), _stream()), 'first_crossing')
    return bracket, flags


def _fp(t):
    return None if t is None else _ptr(t, 'loss tensor')


def _bp(t):
    assert t.is_cuda and t.dtype == torch.bool and t.is_contiguous()
    return t.data_ptr()


def stage2_loss_fwd(rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j, mask_a, mask_b, l2, inv_denom, weight,
                    count_dev=None):
    """Seven floats on the device: the six loss terms and their weighted total (psn_stage2_loss_fwd)."""
    N = mask_a.numel()
    out = torch.empty(7, device=mask_a.device, dtype=torch.float32)
    partial = workspace(2048 * 6, mask_a.device)
    _check(_lib.psn_stage2_loss_fwd(_fp(rgb), _fp(rgb_gt), 0 if rgb is None else rgb.shape[0], _fp(alb), _fp(alb_j), _fp(wgt), _fp(wgt_j),
                                    0 if wgt is None else wgt.shape[-1], _fp(vis), _fp(vis_gt), 0 if vis is None else vis.shape[0],
                                    _fp(nrm), _fp(nrm_gt), _fp(nrm_j), _bp(mask_a), _bp(mask_b), N, int(l2),
                                    ctypes.cast((ctypes.c_float * 6)(*[float(x) for x in inv_denom]), ctypes.c_void_p),
                                    ctypes.cast((ctypes.c_float * 6)(*[float(x) for x in weight]), ctypes.c_void_p),
                                    _ptr(count_dev, 'count_dev', True), partial.data_ptr(), out.data_ptr(), _stream()), 'stage2_loss_fwd')
    return out


def stage2_loss_bwd(g_total, rgb, rgb_gt, k_rgb, alb, alb_j, k_alb, wgt, wgt_j, k_wgt, vis, vis_gt, k_vis, nrm, nrm_gt, nrm_j, k_nrm,
                    k_nrmj, mask_a, mask_b, l2, need, count_dev=None):
    """Gradients of the weighted total with respect to the tensors named in ``need`` (a set of 'rgb', 'alb', 'wgt', 'vis',
    'nrm'); returns dict name -> gradient (alb / wgt / nrm also give the jitter gradients as name + '_j')."""
    N = mask_a.numel()
    new = lambda t: torch.empty_like(t)
    d = {}
    if 'rgb' in need: d['rgb'] = new(rgb)
    if 'alb' in need: d['alb'], d['alb_j'] = new(alb), new(alb_j)
    if 'wgt' in need: d['wgt'], d['wgt_j'] = new(wgt), new(wgt_j)
    if 'vis' in need: d['vis'] = new(vis)
    if 'nrm' in need:
        d['nrm'] = new(nrm)
        if nrm_j is not None:
            d['nrm_j'] = new(nrm_j)
    g = lambda k: None if k not in d else d[k].data_ptr()
    _check(_lib.psn_stage2_loss_bwd(_ptr(g_total, 'g_total'), _fp(rgb), _fp(rgb_gt), 0 if rgb is None else rgb.shape[0], float(k_rgb), g('rgb'),
                                    _fp(alb), _fp(alb_j), float(k_alb), g('alb'), g('alb_j'), _fp(wgt), _fp(wgt_j),
                                    0 if wgt is None else wgt.shape[-1], float(k_wgt), g('wgt'), g('wgt_j'), _fp(vis), _fp(vis_gt),
                                    0 if vis is None else vis.shape[0], float(k_vis), g('vis'), _fp(nrm), _fp(nrm_gt), _fp(nrm_j),
                                    float(k_nrm), float(k_nrmj), g('nrm'), g('nrm_j'), _bp(mask_a), _bp(mask_b), N, int(l2),
                                    _ptr(count_dev, 'count_dev', True), _stream()), 'stage2_loss_bwd')
    return d


def pair_sums(x, V, Ns):
    """x [V * Ns, C] (light-major rows) -> (sum over V [Ns, C], sum over Ns [V, C]) in one pass over x (psn_pair_sums)."""
    C = x.shape[1]
    sx = torch.empty(Ns, C, device=x.device, dtype=torch.float32)
    part = torch.empty(2048, V, C, device=x.device, dtype=torch.float32)
    n_chunks = ctypes.c_int(0)
    _check(_lib.psn_pair_sums(_ptr(x, 'x'), V, Ns, C, sx.data_ptr(), part.data_ptr(), ctypes.byref(n_chunks), _stream()), 'pair_sums')
    return sx, part[:n_chunks.value].sum(0)


def pair_sums_group(xs, V, Ns, pe_l, n_pe, want_bias):
    """xs: list of d z [V * Ns, C] (light-major rows) of the layers that read the input block [table(x_n) | table(l_v)]; pe_l [V, >= n_pe]
    the light table.  -> [(sx [Ns, C], dWl [C, 64] with the first n_pe columns written, bias [C] or None)] in two launches
    (psn_pair_sums_group)."""
    C = xs[0].shape[1]
    dev = xs[0].device
    assert 1 <= len(xs) <= 4 and all(x.shape == (V * Ns, C) and x.is_contiguous() for x in xs) and pe_l.stride(1) == 1 and pe_l.shape[0] == V
    ws = workspace(int(_lib.psn_pair_sums_group_workspace(len(xs), V, Ns, C)), dev)
    arr = (PsnPairSumsItem * len(xs))()
    out = []
    for e, x, wb in zip(arr, xs, want_bias):
        sx = torch.empty(Ns, C, device=dev, dtype=torch.float32)
        dWl = torch.empty(C, 64, device=dev, dtype=torch.float32)  # (columns >= n_pe are never read)
        b = torch.empty(C, device=dev, dtype=torch.float32) if wb else None
        e.x, e.sx, e.dWl, e.bias = _ptr(x, 'x'), sx.data_ptr(), dWl.data_ptr(), None if b is None else b.data_ptr()
        out.append((sx, dWl, b))
    _check(_lib.psn_pair_sums_group(len(xs), ctypes.addressof(arr), V, Ns, C, _ptr(pe_l, 'pe_l'), pe_l.stride(0), int(n_pe), 64,
                                    ws.data_ptr(), _stream()), 'pair_sums_group')
    return out


def row_adam(items, idx, step_sizes_dev=None):
    """items: list of (param, grad, exp_avg, exp_avg_sq, beta1, beta2, eps, step_size) with dense [rows, cols] fp32 device
    tensors; idx: int64 device tensor of the rows that move (psn_row_adam).  step_sizes_dev: fp32 device tensor [len(items)]
    the kernel reads the step sizes from instead (psn_row_adam_dev: graph replay)."""
    arr = (PsnRowAdamItem * len(items))()
    for e, (p, g, m, v, b1, b2, eps, ss) in zip(arr, items):
        p2 = p.view(p.shape[0], -1)
        e.param, e.grad, e.exp_avg, e.exp_avg_sq = _ptr(p, 'param'), _ptr(g, 'grad'), _ptr(m, 'exp_avg'), _ptr(v, 'exp_avg_sq')
        e.rows, e.cols, e.one_minus_beta1, e.one_minus_beta2, e.eps, e.step_size = p2.shape[0], p2.shape[1], 1 - b1, 1 - b2, eps, ss
    assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()
    if step_sizes_dev is not None:
        assert step_sizes_dev.is_cuda and step_sizes_dev.dtype == torch.float32 and step_sizes_dev.numel() >= len(items)
        _check(_lib.psn_row_adam_dev(len(items), ctypes.addressof(arr), idx.data_ptr(), idx.numel(), step_sizes_dev.data_ptr(),
                                     _stream()), 'row_adam_dev')
        return
    _check(_lib.psn_row_adam(len(items), ctypes.addressof(arr), idx.data_ptr(), idx.numel(), _stream()), 'row_adam')


def normalize_rows_fwd(x, eps=1e-12):
    """F.normalize(x, dim=-1) for [n, 3] rows in one launch."""
    assert x.dim() == 2 and x.shape[1] == 3 and x.is_contiguous()
    y = torch.empty_like(x)
    _check(_lib.psn_normalize_rows_fwd(_ptr(x, 'x'), x.shape[0], float(eps), y.data_ptr(), _stream()), 'normalize_rows_fwd')
    return y


def normalize_rows_bwd(x, g, eps=1e-12):
    assert x.shape == g.shape and x.shape[1] == 3 and x.is_contiguous() and g.is_contiguous()
    dx = torch.empty_like(x)
    _check(_lib.psn_normalize_rows_bwd(_ptr(x, 'x'), _ptr(g, 'g'), x.shape[0], float(eps), dx.data_ptr(), _stream()), 'normalize_rows_bwd')
    return dx


def light_rows_fwd(dir_table, int_table, idx, eps=1e-12):
    """(normalize(dir_table[idx]) [L, 3], int_table[idx] [L, 1] or None) in one launch (stage2/trainer.py:376-379)."""
    L = idx.shape[0]
    assert dir_table.dim() == 2 and dir_table.shape[1] == 3 and dir_table.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    d = torch.empty(L, 3, device=dir_table.device, dtype=torch.float32)
    it = None
    if int_table is not None:
        assert int_table.shape == (dir_table.shape[0], 1) and int_table.is_contiguous()
        it = torch.empty(L, 1, device=dir_table.device, dtype=torch.float32)
    _check(_lib.psn_light_rows_fwd(_ptr(dir_table, 'dir_table'), _ptr(int_table, 'int_table', True), _iptr(idx, 'idx'), L, float(eps),
                                   d.data_ptr(), None if it is None else it.data_ptr(), _stream()), 'light_rows_fwd')
    return d, it


def light_rows_bwd(dir_table, idx, g_dir, g_int, eps=1e-12):
    """Dense table gradients ([n, 3] or None, [n, 1] or None) of light_rows_fwd, one launch, no separate zero fill."""
    n = dir_table.shape[0]
    dd = torch.empty(n, 3, device=dir_table.device, dtype=torch.float32) if g_dir is not None else None
    di = torch.empty(n, 1, device=dir_table.device, dtype=torch.float32) if g_int is not None else None
    for t in (g_dir, g_int):
        assert t is None or t.is_contiguous()
    _check(_lib.psn_light_rows_bwd(_ptr(dir_table, 'dir_table'), _iptr(idx, 'idx'), idx.shape[0], n, float(eps), _ptr(g_dir, 'g_dir', True),
                                   _ptr(g_int, 'g_int', True), None if dd is None else dd.data_ptr(), None if di is None else di.data_ptr(),
                                   _stream()), 'light_rows_bwd')
    return dd, di


def camera_rays(uv, pose, intrinsics, idx=None, scale=1.0):
    """Normalised camera rays (rend_util.py:90-147, 4 x 4 pose) of the pixels idx (all when None) -> [n, 3], times scale."""
    assert uv.dim() == 3 and uv.shape[0] == 1 and uv.shape[2] == 2 and uv.is_contiguous()
    assert pose.shape == (1, 4, 4) and intrinsics.shape[0] == 1 and intrinsics.shape[1:] == (4, 4) and pose.is_contiguous() and intrinsics.is_contiguous()
    n = uv.shape[1] if idx is None else idx.shape[0]
    out = torch.empty(n, 3, device=uv.device, dtype=torch.float32)
    _check(_lib.psn_camera_rays(_ptr(uv, 'uv'), _ptr(pose, 'pose'), _ptr(intrinsics, 'intrinsics'), _iptr(idx, 'idx', True), n, float(scale),
                                out.data_ptr(), _stream()), 'camera_rays')
    return out


def _w4(weights):
    return ctypes.cast((ctypes.c_float * 4)(*[float(x) for x in weights]), ctypes.c_void_p)


def stage1_loss_fwd(rgb, rgb_gt, diff, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, n_rays, weights, finish=True):
    """(sums [8], terms [5] or None) of psn_stage1_loss_fwd; ``finish=False`` stops after the sums (data parallelism: the
    caller all-reduces sums[4:7] and calls stage1_loss_terms)."""
    N = rgb.shape[0]
    dev = rgb.device
    sums = torch.empty(8, device=dev, dtype=torch.float32)
    terms = torch.empty(5, device=dev, dtype=torch.float32) if finish else None
    partial = workspace(_lib.psn_stage1_loss_partial_floats(), dev)
    _check(_lib.psn_stage1_loss_fwd(_ptr(rgb, 'rgb'), _ptr(rgb_gt, 'rgb_gt'), _fp(diff), _bp(hit) if hit is not None else None, _fp(normal),
                                    _fp(normal_gt), _bp(norm_mask) if norm_mask is not None else None, _fp(acc), _fp(mask_gt),
                                    _bp(mask_valid) if mask_valid is not None else None, N, int(n_rays), _w4(weights), partial.data_ptr(),
                                    sums.data_ptr(), None if terms is None else terms.data_ptr(), _stream()), 'stage1_loss_fwd')
    return sums, terms


def stage1_loss_terms(sums, n_rays, weights, has_grad, has_norm, has_mask):
    terms = torch.empty(5, device=sums.device, dtype=torch.float32)
    _check(_lib.psn_stage1_loss_terms(_ptr(sums, 'sums'), int(n_rays), _w4(weights), int(has_grad), int(has_norm), int(has_mask),
                                      terms.data_ptr(), _stream()), 'stage1_loss_terms')
    return terms


def stage1_loss_bwd(g_loss, sums, rgb, rgb_gt, hit, normal, normal_gt, norm_mask, acc, mask_gt, mask_valid, n_rays, weights, need):
    """Gradients named in ``need`` (subset of 'rgb', 'diff', 'normal', 'acc') -> dict."""
    N = rgb.shape[0]
    dev = rgb.device
    d = {}
    if 'rgb' in need: d['rgb'] = torch.empty_like(rgb)
    if 'diff' in need: d['diff'] = torch.empty(N, device=dev, dtype=torch.float32)
    if 'normal' in need: d['normal'] = torch.empty_like(normal)
    if 'acc' in need: d['acc'] = torch.empty_like(acc)
    g = lambda k: None if k not in d else d[k].data_ptr()
    b = lambda t: None if t is None else _bp(t)
    _check(_lib.psn_stage1_loss_bwd(_ptr(g_loss, 'g_loss'), _ptr(sums, 'sums'), _ptr(rgb, 'rgb'), _ptr(rgb_gt, 'rgb_gt'), b(hit), _fp(normal),
                                    _fp(normal_gt), b(norm_mask), _fp(acc), _fp(mask_gt), b(mask_valid), N, int(n_rays), _w4(weights),
                                    g('rgb'), g('diff'), g('normal'), g('acc'), _stream()), 'stage1_loss_bwd')
    return d


def surface_normals_fwd(g, hit, eps=1e-5):
    """(norm_pred [N, 3], diff [N]) from g [2 N, 3] and the hit flags [N] bool (psn_surface_normals_fwd)."""
    N = hit.shape[0]
    assert g.shape == (2 * N, 3)
    norm_pred = torch.empty(N, 3, device=g.device, dtype=torch.float32)
    diff = torch.empty(N, device=g.device, dtype=torch.float32)
    _check(_lib.psn_surface_normals_fwd(_ptr(g, 'g'), _bp(hit), N, float(eps), norm_pred.data_ptr(), diff.data_ptr(), _stream()),
           'surface_normals_fwd')
    return norm_pred, diff


def surface_normals_bwd(g, hit, d_norm_pred, d_diff, eps=1e-5):
    N = hit.shape[0]
    dg = torch.empty_like(g)
    _check(_lib.psn_surface_normals_bwd(_ptr(g, 'g'), _bp(hit), N, float(eps), _ptr(d_norm_pred, 'd_norm_pred', True),
                                        _ptr(d_diff, 'd_diff', True), dg.data_ptr(), _stream()), 'surface_normals_bwd')
    return dg


def stage1_rays(pix, camera_mat, world_mat, radius):
    """Camera origins [n, 3], normalised ray directions [n, 3] and sphere exit depths [n] of the pixels pix [n, 2]
    (stage1/model/common.py:205-226, rendering.py:576-596), one launch; camera_mat [3|4, 3|4], world_mat [4, 4] on the device."""
    assert pix.dim() == 2 and pix.shape[1] == 2 and world_mat.shape == (4, 4) and camera_mat.shape in ((3, 3), (4, 4))
    assert world_mat.is_contiguous() and camera_mat.is_contiguous()
    n = pix.shape[0]
    cam, rays = (torch.empty(n, 3, device=pix.device, dtype=torch.float32) for _ in range(2))
    far = torch.empty(n, device=pix.device, dtype=torch.float32)
    _check(_lib.psn_stage1_rays(_ptr(pix, 'pix'), _ptr(camera_mat, 'camera_mat'), camera_mat.shape[0], _ptr(world_mat, 'world_mat'),
                                float(radius ** 2), n, cam.data_ptr(), rays.data_ptr(), far.data_ptr(), _stream()), 'stage1_rays')
    return cam, rays, far


def surface_points(d_pred, flags, cam, rays, want_d=False):
    """(dists [n], obj_mask [n] bool, points [n, 3][, d_i [n]]) from the root finder's depths and the crossing flags
    (stage1/model/rendering.py:516-522, 84-108), one launch."""
    n = d_pred.shape[0]
    assert flags.dtype == torch.int32 and flags.shape == (n,) and flags.is_contiguous() and cam.shape == (n, 3) and rays.shape == (n, 3)
    dev = d_pred.device
    dists = torch.empty(n, device=dev, dtype=torch.float32)
    obj = torch.empty(n, device=dev, dtype=torch.bool)
    pts = torch.empty(n, 3, device=dev, dtype=torch.float32)
    d_i = torch.empty(n, device=dev, dtype=torch.float32) if want_d else None
    _check(_lib.psn_surface_points(_ptr(d_pred, 'd_pred'), flags.data_ptr(), _ptr(cam, 'cam'), _ptr(rays, 'rays'), n,
                                   None if d_i is None else d_i.data_ptr(), dists.data_ptr(), obj.data_ptr(), pts.data_ptr(),
                                   _stream()), 'surface_points')
    return (dists, obj, pts, d_i) if want_d else (dists, obj, pts)


def stage1_targets(pix, img, mask=None, mask_valid=None, normal=None, norm_mask=None, world_mat=None, cos_thresh=None,
                   want_normal=False):
    """Ground truth of the sampled pixels pix [n, 2] (stage1/model/common.py:172-202, training.py:166-191), one launch:
    (rgb_gt [n, 3], mask_gt [n] float, mask_valid [n] bool, norm_mask_gt [n] bool or None, normal_gt [n, 3] or None)."""
    n = pix.shape[0]
    assert pix.dim() == 2 and pix.shape[1] == 2 and img.dim() == 3 and img.shape[0] == 3
    h, w = int(img.shape[1]), int(img.shape[2])
    for t in (mask, mask_valid, norm_mask):
        assert t is None or tuple(t.shape) == (h, w)
    assert normal is None or tuple(normal.shape) == (3, h, w)
    dev = pix.device
    rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
    mask_gt = torch.empty(n, device=dev, dtype=torch.float32)
    valid = torch.empty(n, device=dev, dtype=torch.bool)
    nmask = torch.empty(n, device=dev, dtype=torch.bool) if norm_mask is not None else None
    ngt = torch.empty(n, 3, device=dev, dtype=torch.float32) if want_normal else None
    _check(_lib.psn_stage1_targets(_ptr(pix, 'pix'), n, h, w, _ptr(img, 'img'), _ptr(mask, 'mask', True), _ptr(mask_valid, 'mask_valid', True),
                                   _ptr(normal, 'normal', True) if want_normal else None, _ptr(norm_mask, 'norm_mask', True),
                                   _ptr(world_mat, 'world_mat', True) if want_normal else None, int(cos_thresh is not None),
                                   float(cos_thresh if cos_thresh is not None else 0.0), rgb.data_ptr(), mask_gt.data_ptr(), valid.data_ptr(),
                                   None if ngt is None else ngt.data_ptr(), None if nmask is None else nmask.data_ptr(), _stream()),
           'stage1_targets')
    return rgb, mask_gt, valid, nmask, ngt


def adam_flat(param, grad, exp_avg, exp_avg_sq, segs, beta1, beta2, eps, scalars_dev=None):
    """segs: [(offset, grad_offset, n, neg_step_size, bias_correction2_sqrt)] ranges of the flat buffers (psn_adam_flat);
    more than ADAM_MAX_SEGS ranges go out in several launches.  scalars_dev: fp32 device tensor [len(segs), 2] that the
    kernel reads (neg_step_size, bias_correction2_sqrt) of every range from instead (psn_adam_flat_dev: graph replay)."""
    if scalars_dev is not None:
        assert scalars_dev.is_cuda and scalars_dev.dtype == torch.float32 and scalars_dev.is_contiguous() and scalars_dev.numel() >= 2 * len(segs)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dim() == 1 and t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
    assert exp_avg.numel() == exp_avg_sq.numel() == param.numel()
    for s0 in range(0, len(segs), ADAM_MAX_SEGS):
        part = segs[s0:s0 + ADAM_MAX_SEGS]
        arr = (PsnAdamSeg * len(part))()
        for a, (off, goff, n, ns, bc) in zip(arr, part):
            assert 0 <= off and off + n <= param.numel() and 0 <= goff and goff + n <= grad.numel()
            a.offset, a.grad_offset, a.n, a.neg_step_size, a.bias_correction2_sqrt = int(off), int(goff), int(n), float(ns), float(bc)
        if scalars_dev is not None:
            _check(_lib.psn_adam_flat_dev(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), len(part), arr,
                                          float(1 - beta1), float(beta2), float(1 - beta2), float(eps),
                                          scalars_dev.data_ptr() + 8 * s0, _stream()), 'adam_flat_dev')
            continue
        _check(_lib.psn_adam_flat(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), len(part), arr,
                                  float(1 - beta1), float(beta2), float(1 - beta2), float(eps), _stream()), 'adam_flat')


def shadow_points(surf, ldir, n_steps, lnear, lfar, u, omu, box):
    """In-box shadow-ray sample points (psn_shadow_points) -> (pts [cap,3], rows [cap] int64, counter [1] int64 on the
    device: the number of valid leading entries).  cap = all L*Ns*n_steps rows (worst case)."""
    L, Ns = ldir.shape[0], surf.shape[0]
    cap = L * Ns * n_steps
    pts = torch.empty(max(cap, 1), 3, device=surf.device, dtype=torch.float32)
    rows = torch.empty(max(cap, 1), device=surf.device, dtype=torch.int64)
    counter = torch.zeros(1, device=surf.device, dtype=torch.int64)
    _check(_lib.psn_shadow_points(_ptr(surf, 'surf'), _ptr(ldir, 'ldir'), Ns, L, int(n_steps), float(lnear), float(lfar),
                                  _ptr(u, 'u'), _ptr(omu, 'omu'), float(box), pts.data_ptr(), rows.data_ptr(), counter.data_ptr(),
                                  _stream()), 'shadow_points')
    return pts, rows, counter


def mlp_infer_pe(desc, packed_w, packed_b, points, pe_octaves, pe_scale, out=None, macs_per_row=None, n_rows_dev=None, out_rows=None):
    """Network on gamma(pe_scale * points) with the encoding formed inside the kernel (psn_mlp_infer_pe) -> [Q, n_out].
    ``n_rows_dev`` (int64 [1] on the device): only the first n_rows_dev[0] points are valid (psn_mlp_infer_pe_indirect; the
    grid covers all Q); ``out_rows`` (int64 [Q]): row r is written to out[out_rows[r]] -- ``out`` is then required."""
    Q = points.shape[0]
    assert points.shape == (Q, 3) and points.is_contiguous()
    indirect = n_rows_dev is not None
    if out is None:
        assert out_rows is None, 'a scatter needs its destination'
        out = torch.empty(Q, desc.n_out, device=points.device, dtype=torch.float32)
    assert out.is_contiguous() and (out_rows is not None or out.numel() == Q * desc.n_out)
    if Q == 0:
        return out
    if indirect:
        assert n_rows_dev.dtype == torch.int64 and n_rows_dev.is_cuda and n_rows_dev.numel() == 1
        assert out_rows is None or (out_rows.dtype == torch.int64 and out_rows.is_cuda and out_rows.is_contiguous() and out_rows.numel() >= Q)
        # (no flop count for the profile: how many rows were evaluated is known to the device only)
        with _Prof('mlp_infer', Q, None):
            _check(_lib.psn_mlp_infer_pe_indirect(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'),
                                                  _ptr(points, 'points'), Q, n_rows_dev.data_ptr(),
                                                  None if out_rows is None else out_rows.data_ptr(), int(pe_octaves), float(pe_scale),
                                                  out.data_ptr(), _stream()), 'mlp_infer_pe_indirect')
        return out
    assert out_rows is None
    with _Prof('mlp_infer', Q, None if macs_per_row is None else 2.0 * macs_per_row * Q):
        _check(_lib.psn_mlp_infer_pe(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'), _ptr(points, 'points'),
                                     Q, int(pe_octaves), float(pe_scale), out.data_ptr(), _stream()), 'mlp_infer_pe')
    return out


def march_sweep(desc, packed_w, packed_b, origin, direction, far, u, omu, near, n_steps, tau, pe_octaves, pe_scale,
                early_exit=True, macs_per_row=None):
    """Occupancy of the n_steps sweep points of every ray (psn_march_sweep) -> (occ [N, n_steps], skip flags [N] int32 or None).
    early_exit: blocks of 64 steps behind a ray's first sign change are not evaluated (their occ entries are uninitialised)."""
    N = origin.shape[0]
    assert origin.shape == (N, 3) and direction.shape == (N, 3) and far.shape == (N,) and u.numel() == n_steps == omu.numel()
    occ = torch.empty(N, n_steps, device=origin.device, dtype=torch.float32)
    skip = torch.zeros(N, device=origin.device, dtype=torch.int32) if early_exit else None
    if N == 0:
        return occ, skip
    Q = N * n_steps
    # measurement runs (PROFILE_EVENTS set) count the 64-step blocks that were really evaluated: the event carries the device
    # counter and the flops of ONE block, so that the reader prices evaluated work, not the N x M rows of the dense sweep
    count = torch.zeros(1, device=origin.device, dtype=torch.int64) if PROFILE_EVENTS is not None else None
    with _Prof('march_sweep', Q, None if (macs_per_row is None or count is None) else (count, 2.0 * macs_per_row * 64)):
        _check(_lib.psn_march_sweep(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'), _ptr(origin, 'origin'),
                                    _ptr(direction, 'direction'), _ptr(far, 'far'), _ptr(u, 'u'), _ptr(omu, 'omu'), float(near), N,
                                    int(n_steps), float(tau), int(pe_octaves), float(pe_scale),
                                    None if skip is None else skip.data_ptr(), None if count is None else count.data_ptr(),
                                    occ.data_ptr(), _stream()), 'march_sweep')
    return occ, skip


def root_find(desc, packed_w, packed_b, origin, direction, bracket, tau, n_iter, pe_octaves, pe_scale):
    """n_iter secant iterations on every ray in one launch (psn_root_find) -> d_pred [N]."""
    N = origin.shape[0]
    assert bracket.shape == (4, N)
    out = torch.empty(N, device=origin.device, dtype=torch.float32)
    with _Prof('root_find', N, None):
        _check(_lib.psn_root_find(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'), _ptr(origin, 'origin'),
                                  _ptr(direction, 'direction'), _ptr(bracket, 'bracket'), N, float(tau), int(n_iter), int(pe_octaves),
                                  float(pe_scale), out.data_ptr(), _stream()), 'root_find')
    return out


def workspace(n_floats, device):
    """Scratch buffer per (device, stream), grown on demand: launches on one stream are ordered, so they may share it;
    two streams (the small stage-2 networks run beside the visibility launch on a side stream) must not.  A buffer that a
    HIP-graph capture has used (``capturing(True)`` ... ``capturing(False)`` around the capture: stage2/graph.py) is never
    freed when a larger one replaces it -- the captured launches hold its address."""
    di = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (di, torch._C._cuda_getCurrentRawStream(di))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n_floats:
        if buf is not None and key in _ws_pinned:
            _ws_retired.append(buf)
            _ws_pinned.discard(key)
        buf = torch.empty(max(n_floats, 1 << 22), device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    if _ws_capturing[0]:
        _ws_pinned.add(key)
    return buf


_ws_capturing = [False]
_ws_pinned = set()
_ws_retired = []


def capturing(on):
    """Tell the binding that the launches issued from now on are being captured into a HIP graph (or no longer are)."""
    _ws_capturing[0] = bool(on)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'matrix operand must be row-major with unit column stride'
    return t.stride(0)


def _mat_ptr(t, name):
    """2-D row-major view (may be a column slice of a wider buffer: ld = stride(0))."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError('%s: must be a float32 HIP tensor' % name)
    return t.data_ptr()


def gemm(A, B, trans_a=False, trans_b=False, bias=None, epi=EPI_NONE, aux_in=None, out=None, aux_out=None,
         split_k=1, aux_in2=None, colsum_a=None):
    """out[M,N] = epi(op(A) @ op(B)).  A/B/out/aux may be row-major views with a row stride.
    colsum_a (trans_a only): [M] tensor that receives the column sums of A (= the bias gradient when A = dZ)."""
    if trans_a:
        K, M = A.shape
    else:
        M, K = A.shape
    if trans_b:
        N, Kb = B.shape
    else:
        Kb, N = B.shape
    assert K == Kb, 'gemm: inner dimensions differ (%d vs %d)' % (K, Kb)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    assert out.shape == (M, N)
    ws = None
    if split_k > 1:
        ws = workspace(split_k * M * (N + 1), A.device)
    _check(_lib.psn_gemm(int(trans_a), int(trans_b), M, N, K, _mat_ptr(A, 'A'), _ld(A), _mat_ptr(B, 'B'), _ld(B),
                         _mat_ptr(out, 'out'), _ld(out), _ptr(bias, 'bias', True), epi,
                         None if aux_in is None else _mat_ptr(aux_in, 'aux_in'), 0 if aux_in is None else _ld(aux_in),
                         None if aux_in2 is None else _mat_ptr(aux_in2, 'aux_in2'),
                         0 if aux_in2 is None else _ld(aux_in2),
                         None if aux_out is None else _mat_ptr(aux_out, 'aux_out'),
                         0 if aux_out is None else _ld(aux_out), split_k,
                         None if ws is None else ws.data_ptr(), _ptr(colsum_a, 'colsum_a', True), _stream()), 'gemm')
    return out


def _tn_aligned(t):
    return t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(0) >= 4)


def _tn_is_big(it):
    """Products that psn_gemm_tn_grouped sends to its one-256x256-tile-per-workgroup kernel."""
    return (not it.get('b_div') and not it.get('b_mod') and 128 < it['A'].shape[1] <= 256 and 128 < it['B'].shape[1] <= 256
            and _tn_aligned(it['A']) and _tn_aligned(it['B']) and _tn_aligned(it.get('A2')) and _tn_aligned(it.get('B2')))


def _tn_is_tall(it):
    """Products that psn_gemm_tn_grouped sends to its 256 x (<= 64)-tile kernel (input-block gradients)."""
    return (not it.get('b_div') and not it.get('b_mod') and 128 < it['A'].shape[1] <= 256 and it['B'].shape[1] <= 64
            and it.get('B_tab2') is None
            and _tn_aligned(it['A']) and _tn_aligned(it['B']) and _tn_aligned(it.get('A2')) and _tn_aligned(it.get('B2')))


# EXPERIMENT (BASELINE configs[4] bf16 path): the 256 x 256-tile weight gradients on the bf16 matrix pipe with split operands
# (psn_gemm_tn_grouped_x3).  Process-wide switch for A/B runs and the labelled bench objects; never on by default.
WGRAD_X3 = os.environ.get('PSN_WGRAD_X3', '0') == '1'


class wgrad_precision(object):
    """``with hip.wgrad_precision(mode):`` -- the 256 x 256-tile weight-gradient products issued inside take
    'fp32'   the exact kernel (v_mfma_f32_32x32x2_f32);
    'bf16x6' the split-bf16 kernel with three pieces per operand and six partial products (fp32-class results);
    'bf16x3' two pieces, three partial products (~16 significant bits);
    'bf16'   plain bf16 operands, fp32 accumulation.
    Wrap the BACKWARD pass (that is where the products are issued)."""
    PRODUCTS = {'bf16x6': 6, 'bf16x3': 3, 'bf16': 1}

    def __init__(self, mode):
        assert mode in ('fp32', 'bf16x6', 'bf16x3', 'bf16'), mode
        self.mode = mode

    def __enter__(self):
        global WGRAD_X3
        self.saved, WGRAD_X3 = WGRAD_X3, self.mode != 'fp32'
        self.saved_products = _lib.psn_gemm_tn_x3_set_products(self.PRODUCTS.get(self.mode, 6))
        return self

    def __exit__(self, *exc):
        global WGRAD_X3
        WGRAD_X3 = self.saved
        _lib.psn_gemm_tn_x3_set_products(self.saved_products)
        return False


def gemm_tn_grouped(items, split_k=None, x3=None):
    """Weight gradients of one backward pass in one launch.  items: list of dicts with A [K,M], B [K,N] (row-major
    views, row stride allowed), optional A2 / B2 (second product summed into the same result), optional out [M,N]
    (+ accumulate=True) and colsum (True -> the column sums of A are returned too).  Returns [(C, colsum or None)].
    split_k = K slices per product of the 128 x 128-tile path (default: enough (tile, slice) work items to cover the
    256 CUs about four times with K chunks of at least 512 rows); the 256 x 256-tile path picks its own."""
    res = []
    K = items[0]['A'].shape[0]
    dev = items[0]['A'].device
    # which kernel of psn_gemm_tn_grouped a product goes to: decided ONCE per item (the predicates were evaluated four times each)
    kinds = {id(it): ('big' if _tn_is_big(it) else ('tall' if _tn_is_tall(it) else 'tile')) for it in items}
    if split_k is None:
        work = 0
        for it in items:
            if kinds[id(it)] == 'tile':
                tiles = ((it['A'].shape[1] + 127) // 128) * ((it['B'].shape[1] + 127) // 128)
                work += tiles * (2 if it.get('A2') is not None else 1)
        want = max(1, (1024 + work - 1) // max(work, 1))
        # K chunks of at least 512 rows -- for a SHORT product (the 128- / 64-wide BRDF / normal networks of a 4096-pixel rank
        # shard: K = 7k rows, 5 - 6 tiles) that bound left 84 workgroups of 33 k-steps each on 256 CUs, 100 us per call that
        # nothing else filled; there a chunk may shrink to 128 rows (8 k-steps): 340 workgroups, the reduction is 20 MB
        min_rows = 512 if K >= 16384 else 128
        split_k = int(max(1, min(want, K // min_rows if K >= min_rows else 1, 256)))
    for c0 in range(0, len(items), MAX_GROUP):
        chunk = items[c0:c0 + MAX_GROUP]
        arr = (PsnGemmTnItem * len(chunk))()
        need = 0
        keep = []
        # slices per product of the one-tile-per-workgroup path (mirrors psn_gemm_tn_grouped, which re-checks the size)
        is_big = lambda it_: kinds[id(it_)] == 'big'
        n_big = sum((2 if it.get('A2') is not None else 1) for it in chunk if is_big(it))
        split_big = 1
        if n_big:
            want = max(1, min(256 // n_big, K // 256 if K >= 256 else 1))
            kc = -(-K // want)
            kc = (kc + 15) // 16 * 16  # (the kernel rounds its K chunk to 32-row k-tiles: never more slices than this bound)
            split_big = -(-K // kc)
        n_tall = sum((2 if it.get('A2') is not None else 1) for it in chunk if kinds[id(it)] == 'tall')
        split_tall = 1
        if n_tall:
            want = max(1, min(256 // n_tall, K // 256 if K >= 256 else 1))
            kc = -(-K // want)
            kc = (kc + 15) // 16 * 16
            split_tall = -(-K // kc)
        for i, it in enumerate(chunk):
            A, B = it['A'], it['B']
            b_div, b_mod = int(it.get('b_div', 0)), int(it.get('b_mod', 0))
            if b_div or b_mod:  # B is a table: row k of the product reads B[(k // b_div) % b_mod]
                b_div, b_mod = max(b_div, 1), (b_mod if b_mod else B.shape[0])
                assert A.shape[0] == K and b_mod <= B.shape[0] and it.get('A2') is None
            else:
                # a product may cover a row PREFIX of the pass (operands with fewer rows than items[0]): k_rows
                assert A.shape[0] == B.shape[0] <= K, 'gemm_tn_grouped: a product has at most the K rows of the first one'
            M, N = A.shape[1], B.shape[1]
            Bt2 = it.get('B_tab2')
            if Bt2 is not None:  # second table side by side: columns N .. N + Bt2.shape[1] - 1 of the virtual operand
                assert N % 4 == 0 and int(it.get('b2_div', 0)) > 0
                N += Bt2.shape[1]
            C = it.get('out')
            if C is None:
                C = torch.empty(M, N, device=dev, dtype=torch.float32)
            assert C.shape == (M, N)
            cs = torch.empty(M, device=dev, dtype=torch.float32) if it.get('colsum') else None
            A2, B2 = it.get('A2'), it.get('B2')
            e = arr[i]
            e.A, e.lda, e.B, e.ldb = _mat_ptr(A, 'A'), _ld(A), _mat_ptr(B, 'B'), _ld(B)
            if A2 is not None:
                assert A2.shape == A.shape and B2.shape == B.shape
                e.A2, e.lda2, e.B2, e.ldb2 = _mat_ptr(A2, 'A2'), _ld(A2), _mat_ptr(B2, 'B2'), _ld(B2)
            e.C, e.ldc, e.M, e.N = _mat_ptr(C, 'out'), _ld(C), M, N
            e.accumulate = int(bool(it.get('accumulate')))
            e.b_div, e.b_mod = b_div, b_mod
            if Bt2 is not None:
                e.B_tab2, e.ldb_tab2 = _mat_ptr(Bt2, 'B_tab2'), _ld(Bt2)
                e.b2_div, e.b2_mod, e.b_split = int(it['b2_div']), int(it.get('b2_mod', Bt2.shape[0])), B.shape[1]
            e.colsum_a = None if cs is None else cs.data_ptr()
            e.k_rows = 0 if A.shape[0] == K else A.shape[0]
            sk = max(split_k, split_big) if is_big(it) else (max(split_k, split_tall) if kinds[id(it)] == 'tall' else split_k)
            need += (2 if A2 is not None else 1) * sk * M * N + sk * M + 16
            keep.append((C, cs))
        ws = workspace(need, dev)
        flops = sum(2.0 * (arr[i].k_rows or K) * arr[i].M * arr[i].N * (2 if chunk[i].get('A2') is not None else 1) for i in range(len(chunk)))
        use_x3 = WGRAD_X3 if x3 is None else bool(x3)
        with _Prof('gemm_tn_grouped', flops):
            _check((_lib.psn_gemm_tn_grouped_x3 if use_x3 else _lib.psn_gemm_tn_grouped)(
                len(chunk), ctypes.addressof(arr), K, split_k, ws.data_ptr(), ws.numel(), _stream()), 'gemm_tn_grouped')
        res += keep
    return res


def colsum(X, out=None, accumulate=False, row_weight=None):
    """Plain column sums [N]: out[n] (+)= sum_m X[m, n]; or with row_weight [M, n_w <= 4] (a 1-D [M] counts as one
    column) the small weight gradient [n_w, N]: out[j, n] (+)= sum_m w[m, j] X[m, n]."""
    M, N = X.shape
    n_w, ldw, wp = 0, 0, None
    if row_weight is not None:
        w2 = row_weight.reshape(M, -1)
        assert w2.stride(1) == 1 and 1 <= w2.shape[1] <= 4
        n_w, ldw, wp = w2.shape[1], w2.stride(0), _mat_ptr(w2, 'row_weight')
    if out is None:
        out = torch.empty((n_w, N) if n_w else (N,), device=X.device, dtype=torch.float32)
    ws = workspace(2048 * N, X.device)
    _check(_lib.psn_colsum(_mat_ptr(X, 'X'), wp, n_w, ldw, M, N, _ld(X), out.data_ptr(), int(accumulate), ws.data_ptr(),
                           _stream()), 'colsum')
    return out


# --------------------------------------------------------------------------- fused MLP inference
def mlp_pack_layer(W, n_mt, k_tiles, dst, transpose=False):
    """W: 2-D fp32 device matrix (row-major view, unit column stride); with transpose=True its transpose is packed.
    Zero-extended to [n_mt*32, k_tiles*32] and written to dst (flat float view) in stage order."""
    assert W.dim() == 2 and W.stride(1) == 1 and W.is_cuda and W.dtype == torch.float32
    rows, cols = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
    _check(_lib.psn_mlp_pack_layer(W.data_ptr(), W.stride(0), rows, cols, int(transpose), n_mt, k_tiles, dst.data_ptr(),
                                   _stream()), 'mlp_pack_layer')


PACK_MAX_ITEMS = 24


def mlp_pack_layers(plan):
    """plan: list of (W, transpose, n_mt, k_tiles, dst[, format = W_F32]) like mlp_pack_layer, packed in ONE launch per 24 blocks."""
    for c0 in range(0, len(plan), PACK_MAX_ITEMS):
        chunk = plan[c0:c0 + PACK_MAX_ITEMS]
        arr = (PsnPackItem * len(chunk))()
        for e, item in zip(arr, chunk):
            W, transpose, n_mt, k_tiles, dst = item[:5]
            e.format = item[5] if len(item) > 5 else W_F32
            assert W.dim() == 2 and W.stride(1) == 1 and W.is_cuda and W.dtype == torch.float32
            assert dst.is_contiguous() and dst.numel() == n_mt * k_tiles * 1024
            rows, cols = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
            e.W, e.dst, e.ldw = W.data_ptr(), dst.data_ptr(), W.stride(0)
            e.rows, e.cols, e.transpose, e.n_mt, e.k_tiles = rows, cols, int(transpose), n_mt, k_tiles
        _check(_lib.psn_mlp_pack_layers(len(chunk), ctypes.addressof(arr), _stream()), 'mlp_pack_layers')


class block_order(object):
    """``with hip.block_order('row'):`` -- the lean engine's (group, point) row sets in row order instead of the default
    point-tile-major order (psn_mlp_block_order; results are bit-identical, the order only decides what stays in L2)."""

    def __init__(self, order):
        assert order in ('point', 'row'), order
        self.order = order

    def __enter__(self):
        self.saved = _lib.psn_mlp_block_order(1 if self.order == 'point' else 0)
        return self

    def __exit__(self, *exc):
        _lib.psn_mlp_block_order(self.saved)
        return False


def mlp_infer(desc, packed_w, packed_b, tab_a, a_div, a_mod, tab_b, b_div, b_mod, n_rows, out=None, init_a=None,
              init_b=None, save=None, save_row0=0, mask=None, aux2=None, save2=None, act_init=None, macs_per_row=None,
              rank_init=None, save_tiles=None, save2_tiles=None, act_init_rows=None, live=None, save_bits=None):
    """save: list (one entry per hidden layer, None allowed) of [n_rows - save_row0, 256] tensors that receive the
    post-activation outputs of the rows >= save_row0.
    rank_init = (coef [n_rows, k], basis [k, init_stride]), k <= 4: rank-k init of the layers with init_off >= 0.
    act_init [act_init_rows (default n_rows), width]: initial activations; the rows behind them start from zeros.
    live = (count float32 [1] on the device, period): the rows [0, save_row0) are groups of `period` rows of which the first
    count[0] are real (psn_mlp_infer_padded; plain forward launches only): all-padding workgroups write zeros and leave.
    save_bits: list like ``save`` of int64 [n_rows - save_row0, 4] tensors (or None) that receive the sign bits of the dumped
    activations (psn_mlp_infer_bits); a chain layer with ACT_RELU_BITS takes such a tensor as its ``mask`` entry."""
    tiles_arr = None
    if save_tiles is not None or save2_tiles is not None:  # per layer: bit mt = the 16-column tile mt of the dump is written
        tiles_arr = (ctypes.c_uint32 * (2 * MAX_LAYERS))(*([0xFFFFFFFF] * (2 * MAX_LAYERS)))
        for base, lst in ((0, save_tiles), (MAX_LAYERS, save2_tiles)):
            for l, m in enumerate(lst or []):
                if m is not None:
                    tiles_arr[base + l] = int(m)
    rk_coef = rk_basis = None
    rk_k = 0
    if rank_init is not None:
        rk_coef, rk_basis = rank_init
        rk_k = rk_basis.shape[0]
        assert rk_coef.shape == (n_rows, rk_k) and rk_coef.is_contiguous() and rk_basis.is_contiguous() and 1 <= rk_k <= 4
        assert rk_basis.shape[1] == desc.init_stride, 'rank_init: the basis rows are init_stride floats'

    ai_rows = 0
    if act_init is not None:
        ai_rows = n_rows if act_init_rows is None else act_init_rows
        assert act_init.is_contiguous() and act_init.shape[0] >= ai_rows, 'act_init: fewer rows than act_init_rows'
    if out is None and desc.n_out > 0:
        out = torch.empty(n_rows, desc.n_out, device=packed_w.device, dtype=torch.float32)
    n_hidden = desc.n_layers - 1 if desc.n_out > 0 else desc.n_layers
    def ptr_array(lst, n, name):
        if lst is None:
            return None
        assert len(lst) == n, '%s: expected %d entries' % (name, n)
        return (ctypes.c_void_p * n)(*[None if t is None else (_bits_ptr(t, name) if t.dtype == torch.int64 else _ptr(t, name)) for t in lst])

    mask_arr = ptr_array(mask, desc.n_layers, 'mask')
    aux2_arr = ptr_array(aux2, desc.n_layers, 'aux2')
    save2_arr = ptr_array(save2, desc.n_layers, 'save2')
    save_arr = None
    if save is not None:
        assert len(save) == n_hidden
        save_arr = (ctypes.c_void_p * len(save))(*[None if t is None else _ptr(t, 'save') for t in save])
    # 'mlp_infer' = the lean engine, 'mlp_chain' = the chain engine (same dispatch rule as psn_mlp_infer)
    chain = act_init is not None or rank_init is not None or tiles_arr is not None or mask is not None or aux2 is not None or save2 is not None or any(
        desc.layers[l].act > ACT_SOFTPLUS100 for l in range(desc.n_layers))
    if save_bits is not None:
        if chain or save is None:
            raise RuntimeError('mlp_infer: sign-bit words go with the activation dumps of a plain forward launch')
        assert len(save_bits) == n_hidden
        for t, b in zip(save, save_bits):
            assert b is None or (t is not None and b.shape == (t.shape[0], 4))
        bits_arr = (ctypes.c_void_p * n_hidden)(*[None if t is None else _bits_ptr(t, 'save_bits') for t in save_bits])
        cnt, period = live if live is not None else (None, 0)
        with _Prof('mlp_infer', n_rows, None if macs_per_row is None else 2.0 * macs_per_row * n_rows):
            _check(_lib.psn_mlp_infer_bits(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'),
                                           _ptr(tab_a, 'tab_a', True), a_div, a_mod, _ptr(tab_b, 'tab_b', True), b_div, b_mod,
                                           _ptr(init_a, 'init_a', True), _ptr(init_b, 'init_b', True), save_arr, bits_arr, save_row0, n_rows,
                                           _ptr(out, 'out', True), None if cnt is None else cnt.data_ptr(), int(period), _stream()), 'mlp_infer_bits')
        return out
    if live is not None:
        cnt, period = live
        if chain:
            raise RuntimeError('mlp_infer: a device-side live count is for plain forward launches (no chain operands)')
        assert cnt.is_cuda and cnt.dtype == torch.float32 and cnt.numel() == 1 and int(period) >= 1
        with _Prof('mlp_infer', n_rows, None if macs_per_row is None else 2.0 * macs_per_row * n_rows):
            _check(_lib.psn_mlp_infer_padded(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'),
                                             _ptr(tab_a, 'tab_a', True), a_div, a_mod, _ptr(tab_b, 'tab_b', True), b_div, b_mod,
                                             _ptr(init_a, 'init_a', True), _ptr(init_b, 'init_b', True), save_arr, save_row0, n_rows,
                                             _ptr(out, 'out', True), cnt.data_ptr(), int(period), _stream()), 'mlp_infer_padded')
        return out
    with _Prof('mlp_chain' if chain else 'mlp_infer', n_rows, None if macs_per_row is None else 2.0 * macs_per_row * n_rows):
        _check(_lib.psn_mlp_infer(ctypes.byref(desc), _ptr(packed_w, 'packed_w'), _ptr(packed_b, 'packed_b'),
                                  _ptr(tab_a, 'tab_a', True), a_div, a_mod, _ptr(tab_b, 'tab_b', True), b_div, b_mod,
                                  _ptr(init_a, 'init_a', True), _ptr(init_b, 'init_b', True), save_arr, save_row0, mask_arr,
                                  aux2_arr, save2_arr, _ptr(act_init, 'act_init', True), int(ai_rows), _ptr(rk_coef, 'rk_coef', True),
                                  _ptr(rk_basis, 'rk_basis', True), rk_k, tiles_arr, n_rows, _ptr(out, 'out', True), _stream()), 'mlp_infer')
    return out


# --------------------------------------------------------------------------- weight normalisation
WN_MAX_ITEMS = 16


def weight_norm_fwd(vs, gs, scales):
    """[v_l * (g_l / |v_l|_row) * scale_l] for all layers in one launch (<= 16 per launch)."""
    out = [torch.empty_like(v) for v in vs]
    for c0 in range(0, len(vs), WN_MAX_ITEMS):
        n = min(WN_MAX_ITEMS, len(vs) - c0)
        arr = (PsnWnItem * n)()
        for i in range(n):
            v, g = vs[c0 + i], gs[c0 + i]
            assert v.dim() == 2 and g.numel() == v.shape[0]
            e = arr[i]
            e.v, e.g, e.w = _ptr(v, 'weight_v'), _ptr(g, 'weight_g'), out[c0 + i].data_ptr()
            e.rows, e.cols, e.scale = v.shape[0], v.shape[1], float(scales[c0 + i])
        _check(_lib.psn_weight_norm_fwd(n, ctypes.addressof(arr), _stream()), 'weight_norm_fwd')
    return out


def weight_norm_bwd(vs, gs, scales, dws):
    """(dv_l, dg_l) of weight_norm_fwd for the output gradients dws (dense, same shapes as vs)."""
    dvs = [torch.empty_like(v) for v in vs]
    dgs = [torch.empty_like(g) for g in gs]
    for c0 in range(0, len(vs), WN_MAX_ITEMS):
        n = min(WN_MAX_ITEMS, len(vs) - c0)
        arr = (PsnWnItem * n)()
        for i in range(n):
            v, g = vs[c0 + i], gs[c0 + i]
            e = arr[i]
            e.v, e.g, e.dw = _ptr(v, 'weight_v'), _ptr(g, 'weight_g'), _ptr(dws[c0 + i], 'dw')
            e.dv, e.dg = dvs[c0 + i].data_ptr(), dgs[c0 + i].data_ptr()
            e.rows, e.cols, e.scale = v.shape[0], v.shape[1], float(scales[c0 + i])
        _check(_lib.psn_weight_norm_bwd(n, ctypes.addressof(arr), _stream()), 'weight_norm_bwd')
    return dvs, dgs


# --------------------------------------------------------------------------- bf16 inference engine (evaluation only)
def mlp_pack_bf16(W, permuted, n_ot, ks0, n_ks, dst):
    """k-steps [ks0, ks0 + n_ks) of the fp32 matrix W (row-major view) -> dst (flat bfloat16 view, n_ks*n_ot*512
    elements) in fragment order of the bf16 engine."""
    assert W.dim() == 2 and W.stride(1) == 1 and W.is_cuda and W.dtype == torch.float32
    assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() == n_ks * n_ot * 512
    _check(_lib.psn_mlp_pack_bf16(W.data_ptr(), W.stride(0), W.shape[0], W.shape[1], int(permuted), n_ot, ks0, n_ks,
                                  dst.data_ptr(), _stream()), 'mlp_pack_bf16')


def _bf16_table(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == 64):
        raise RuntimeError('%s: must be a contiguous [n, 64] bfloat16 HIP tensor' % name)
    return t.data_ptr()


def mlp_infer_bf16(desc, packed_w, final_bias, tab_a, a_div, a_mod, tab_b, b_div, b_mod, n_rows, out=None):
    if out is None:
        out = torch.empty(n_rows, desc.n_out, device=packed_w.device, dtype=torch.float32)
    assert packed_w.dtype == torch.bfloat16 and packed_w.is_cuda and packed_w.is_contiguous()
    assert final_bias.numel() == 32
    with _Prof('mlp_infer_bf16', n_rows):
        _check(_lib.psn_mlp_infer_bf16(ctypes.byref(desc), packed_w.data_ptr(), _ptr(final_bias, 'final_bias'),
                                       _bf16_table(tab_a, 'tab_a'), a_div, a_mod, _bf16_table(tab_b, 'tab_b'), b_div, b_mod,
                                       n_rows, _ptr(out, 'out'), _stream()), 'mlp_infer_bf16')
    return out


def bf16_pack_group_bias(V, dst=None):
    """V [n, 256] fp32 (one row per (group, input layer)) -> [n, 4096] bfloat16 bias k-steps of the bf16 engine."""
    assert V.is_cuda and V.dtype == torch.float32 and V.is_contiguous() and V.dim() == 2 and V.shape[1] == 256
    if dst is None:
        dst = torch.empty(V.shape[0], 4096, device=V.device, dtype=torch.bfloat16)
    assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() == V.shape[0] * 4096
    _check(_lib.psn_bf16_pack_group_bias(V.data_ptr(), V.shape[0], dst.data_ptr(), _stream()), 'bf16_pack_group_bias')
    return dst


def mlp_infer_bf16_grouped(desc, packed_w, final_bias, tab_a, group_bias, n_groups, out=None):
    """Rows (g, n) -> g * tab_a.shape[0] + n; group_bias [n_groups * n_in_layers, 4096] bfloat16 (bf16_pack_group_bias)."""
    rows = tab_a.shape[0]
    if out is None:
        out = torch.empty(n_groups * rows, desc.n_out, device=packed_w.device, dtype=torch.float32)
    assert packed_w.dtype == torch.bfloat16 and packed_w.is_cuda and packed_w.is_contiguous()
    assert final_bias.numel() == 32
    n_in = sum(1 for l in range(desc.n_hidden) if desc.has_in[l])
    if not (group_bias.is_cuda and group_bias.dtype == torch.bfloat16 and group_bias.is_contiguous()
            and group_bias.numel() == n_groups * n_in * 4096):
        raise RuntimeError('group_bias: must be a contiguous [n_groups * %d, 4096] bfloat16 HIP tensor' % n_in)
    assert out.numel() == n_groups * rows * desc.n_out
    with _Prof('mlp_infer_bf16', n_groups * rows):
        _check(_lib.psn_mlp_infer_bf16_grouped(ctypes.byref(desc), packed_w.data_ptr(), _ptr(final_bias, 'final_bias'),
                                               _bf16_table(tab_a, 'tab_a'), rows, group_bias.data_ptr(), n_groups,
                                               _ptr(out, 'out'), _stream()), 'mlp_infer_bf16_grouped')
    return out


# --------------------------------------------------------------------------- split-bf16 ("bf16x6") engine, experiment
X3_KS = 8 * 3 * 512  # bf16 elements of one hidden-layer k-step: 8 output tiles x 3 planes x 1 KB


def x3_pack(W, permuted, n_ot, ks0, n_ks, dst):
    """k-steps [ks0, ks0 + n_ks) of the fp32 matrix W -> dst (flat bfloat16 view, n_ks * n_ot * 3 * 512 elements): the three
    bf16 planes of every weight in fragment order of the split engine (psn_x3_pack)."""
    assert W.dim() == 2 and W.stride(1) == 1 and W.is_cuda and W.dtype == torch.float32
    assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() == n_ks * n_ot * 3 * 512
    _check(_lib.psn_x3_pack(W.data_ptr(), W.stride(0), W.shape[0], W.shape[1], int(permuted), n_ot, ks0, n_ks, dst.data_ptr(), _stream()),
           'x3_pack')


def x3_pack_bias(V, dst=None):
    """V [n, 256] fp32 -> [n, 4096] bfloat16 bias k-steps (K slots 0..2 = hi / mid / lo)."""
    assert V.is_cuda and V.dtype == torch.float32 and V.is_contiguous() and V.dim() == 2 and V.shape[1] == 256
    if dst is None:
        dst = torch.empty(V.shape[0], 4096, device=V.device, dtype=torch.bfloat16)
    assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() == V.shape[0] * 4096
    _check(_lib.psn_x3_pack_bias(V.data_ptr(), V.shape[0], dst.data_ptr(), _stream()), 'x3_pack_bias')
    return dst


def mlp_infer_x3_grouped(desc, packed_w, bias_steps, final_bias, U, V, out=None, macs_per_row=None):
    """Rows (g, n) -> g * U.shape[0] + n on the split-bf16 engine (psn_mlp_infer_x3_grouped); U [Ns, n_in * 256], V [G, n_in * 256]
    fp32 init tables of the layers that read the input block."""
    rows, n_groups = U.shape[0], V.shape[0]
    if out is None:
        out = torch.empty(n_groups * rows, desc.n_out, device=packed_w.device, dtype=torch.float32)
    n_in = sum(1 for l in range(desc.n_hidden) if desc.has_in[l])
    for t, nm in ((packed_w, 'packed_w'), (bias_steps, 'bias_steps')):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
            raise RuntimeError('%s: must be a contiguous bfloat16 HIP tensor' % nm)
    assert U.shape[1] == V.shape[1] == n_in * 256 and bias_steps.numel() == desc.n_hidden * 4096
    assert final_bias.numel() == 32 and out.numel() == n_groups * rows * desc.n_out
    with _Prof('mlp_infer_x3', n_groups * rows, None if macs_per_row is None else 2.0 * macs_per_row * n_groups * rows):
        _check(_lib.psn_mlp_infer_x3_grouped(ctypes.byref(desc), packed_w.data_ptr(), bias_steps.data_ptr(), _ptr(final_bias, 'final_bias'),
                                             _ptr(U, 'U'), rows, _ptr(V, 'V'), n_groups, _ptr(out, 'out'), _stream()), 'mlp_infer_x3_grouped')
    return out


def mlp_infer_x3_occ(desc, packed_w, bias_steps, final_bias, points, pe_octaves, pe_scale, skip_layer, pe_first, out=None,
                     n_rows_dev=None, out_rows=None, macs_per_row=None):
    """sigmoid(-10 logit) of the stage-1 occupancy network for [Q, 3] points on the split-bf16 engine (psn_mlp_infer_x3_occ):
    the encoding is formed in the kernel, the activation is softplus(beta = 100).  n_rows_dev / out_rows as hip.mlp_infer_pe."""
    Q = points.shape[0]
    assert points.dim() == 2 and points.shape[1] == 3
    if out is None:
        assert out_rows is None
        out = torch.empty(Q, desc.n_out, device=points.device, dtype=torch.float32)
    for t, nm in ((packed_w, 'packed_w'), (bias_steps, 'bias_steps')):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
            raise RuntimeError('%s: must be a contiguous bfloat16 HIP tensor' % nm)
    assert bias_steps.numel() == desc.n_hidden * 4096 and final_bias.numel() == 32 and desc.n_out == 1
    if n_rows_dev is not None:
        assert n_rows_dev.is_cuda and n_rows_dev.dtype == torch.int64 and n_rows_dev.numel() == 1
    if out_rows is not None:
        assert out_rows.is_cuda and out_rows.dtype == torch.int64 and out_rows.is_contiguous() and out_rows.numel() >= Q
    with _Prof('mlp_infer_x3_occ', Q, None if macs_per_row is None else 2.0 * macs_per_row * Q):
        _check(_lib.psn_mlp_infer_x3_occ(ctypes.byref(desc), packed_w.data_ptr(), bias_steps.data_ptr(), _ptr(final_bias, 'final_bias'),
                                         _ptr(points, 'points'), Q, None if n_rows_dev is None else n_rows_dev.data_ptr(),
                                         None if out_rows is None else out_rows.data_ptr(), int(pe_octaves), float(pe_scale),
                                         int(skip_layer), int(pe_first), _ptr(out, 'out'), _stream()), 'mlp_infer_x3_occ')
    return out


# --------------------------------------------------------------------------- SG shading
def sg_shade_fwd(light_dir, view, normal, albedo, weights, lobe, light_int, light_int_scalar, vis, specular_rgb):
    L, Ns, nb = light_dir.shape[0], view.shape[0], lobe.shape[0]
    rgb = torch.empty(L * Ns, 3, device=view.device, dtype=torch.float32)
    spec = torch.empty(L * Ns, 3 if specular_rgb else 1, device=view.device, dtype=torch.float32)
    _check(_lib.psn_sg_shade_fwd(_ptr(light_dir, 'light_dir'), _ptr(view, 'view'), _ptr(normal, 'normal'),
                                 _ptr(albedo, 'albedo'), _ptr(weights, 'weights'), _ptr(lobe, 'lobe'),
                                 _ptr(light_int, 'light_int', True),
                                 1 if light_int is None or light_int.dim() == 1 else light_int.shape[1],
                                 float(light_int_scalar), _ptr(vis, 'vis', True),
                                 L, Ns, nb, int(bool(specular_rgb)), _ptr(rgb, 'rgb'), _ptr(spec, 'spec'), _stream()),
           'sg_shade_fwd')
    return rgb, spec


def sg_shade_bwd(light_dir, view, normal, albedo, weights, lobe, light_int, light_int_scalar, vis, specular_rgb,
                 g_rgb, g_spec, want_vis):
    L, Ns, nb = light_dir.shape[0], view.shape[0], lobe.shape[0]
    dev = view.device
    d_albedo = torch.empty(Ns, 3, device=dev)
    d_weights = torch.empty_like(weights)
    d_normal = torch.empty(Ns, 3, device=dev)
    d_vis = torch.empty(L * Ns, device=dev) if want_vis else None
    d_ldir = torch.empty(L, 3, device=dev)
    d_lint = torch.empty(L, device=dev) if light_int is not None else None
    ws = workspace(((Ns + 63) // 64) * L * 4, dev)
    _check(_lib.psn_sg_shade_bwd(_ptr(light_dir, 'light_dir'), _ptr(view, 'view'), _ptr(normal, 'normal'),
                                 _ptr(albedo, 'albedo'), _ptr(weights, 'weights'), _ptr(lobe, 'lobe'),
                                 _ptr(light_int, 'light_int', True), float(light_int_scalar), _ptr(vis, 'vis', True),
                                 L, Ns, nb, int(bool(specular_rgb)), _ptr(g_rgb, 'g_rgb'), _ptr(g_spec, 'g_spec', True),
                                 _ptr(d_albedo, 'd_albedo'), _ptr(d_weights, 'd_weights'), _ptr(d_normal, 'd_normal'),
                                 _ptr(d_vis, 'd_vis', True), _ptr(d_ldir, 'd_ldir'), _ptr(d_lint, 'd_lint', True),
                                 ws.data_ptr(), _stream()), 'sg_shade_bwd')
    return d_albedo, d_weights, d_normal, d_vis, d_ldir, d_lint


# --------------------------------------------------------------------------- GGX microfacet shading
def mf_shade_fwd(light_dir, view, normal, albedo, rough, light_int, light_int_scalar, f0, vis):
    L, Ns = light_dir.shape[0], view.shape[0]
    rgb = torch.empty(L * Ns, 3, device=view.device, dtype=torch.float32)
    _check(_lib.psn_mf_shade_fwd(_ptr(light_dir, 'light_dir'), _ptr(view, 'view'), _ptr(normal, 'normal'),
                                 _ptr(albedo, 'albedo'), _ptr(rough, 'rough'), _ptr(light_int, 'light_int', True),
                                 float(light_int_scalar), float(f0), _ptr(vis, 'vis', True), L, Ns, _ptr(rgb, 'rgb'),
                                 _stream()), 'mf_shade_fwd')
    return rgb


def mf_shade_bwd(light_dir, view, normal, albedo, rough, light_int, light_int_scalar, f0, vis, g_rgb, want_vis):
    L, Ns = light_dir.shape[0], view.shape[0]
    dev = view.device
    d_albedo = torch.empty(Ns, 3, device=dev)
    d_rough = torch.empty(Ns, device=dev)
    d_normal = torch.empty(Ns, 3, device=dev)
    d_vis = torch.empty(L * Ns, device=dev) if want_vis else None
    d_ldir = torch.empty(L, 3, device=dev)
    d_lint = torch.empty(L, device=dev) if light_int is not None else None
    ws = workspace(((Ns + 63) // 64) * L * 4, dev)
    _check(_lib.psn_mf_shade_bwd(_ptr(light_dir, 'light_dir'), _ptr(view, 'view'), _ptr(normal, 'normal'),
                                 _ptr(albedo, 'albedo'), _ptr(rough, 'rough'), _ptr(light_int, 'light_int', True),
                                 float(light_int_scalar), float(f0), _ptr(vis, 'vis', True), L, Ns, _ptr(g_rgb, 'g_rgb'),
                                 _ptr(d_albedo, 'd_albedo'), _ptr(d_rough, 'd_rough'), _ptr(d_normal, 'd_normal'),
                                 _ptr(d_vis, 'd_vis', True), _ptr(d_ldir, 'd_ldir'), _ptr(d_lint, 'd_lint', True),
                                 ws.data_ptr(), _stream()), 'mf_shade_bwd')
    return d_albedo, d_rough, d_normal, d_vis, d_ldir, d_lint


def march_sweep_x3(desc, packed_w, bias_steps, final_bias, origin, direction, far, u, omu, near, n_steps, tau, pe_octaves, pe_scale,
                   skip_layer, pe_first, early_exit=True, macs_per_row=None):
    """hip.march_sweep on the split-bf16 engine (psn_march_sweep_x3; opt-in experiment): occupancy of the n_steps sweep points of
    every ray -> (occ [N, n_steps], skip flags [N] int32 or None); blocks of 128 steps behind a ray's first sign change are not
    evaluated when early_exit (their entries are uninitialised).  n_steps a multiple of 128."""
    N = origin.shape[0]
    assert origin.shape == (N, 3) and direction.shape == (N, 3) and far.shape == (N,) and u.numel() == n_steps == omu.numel()
    for t, nm in ((packed_w, 'packed_w'), (bias_steps, 'bias_steps')):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
            raise RuntimeError('%s: must be a contiguous bfloat16 HIP tensor' % nm)
    occ = torch.empty(N, n_steps, device=origin.device, dtype=torch.float32)
    skip = torch.zeros(N, device=origin.device, dtype=torch.int32) if early_exit else None
    if N == 0:
        return occ, skip
    count = torch.zeros(1, device=origin.device, dtype=torch.int64) if PROFILE_EVENTS is not None else None
    with _Prof('march_sweep_x3', N * n_steps, None if (macs_per_row is None or count is None) else (count, 2.0 * macs_per_row * 128)):
        _check(_lib.psn_march_sweep_x3(ctypes.byref(desc), packed_w.data_ptr(), bias_steps.data_ptr(), _ptr(final_bias, 'final_bias'),
                                       _ptr(origin, 'origin'), _ptr(direction, 'direction'), _ptr(far, 'far'), _ptr(u, 'u'), _ptr(omu, 'omu'),
                                       float(near), N, int(n_steps), float(tau), int(pe_octaves), float(pe_scale), int(skip_layer), int(pe_first),
                                       None if skip is None else skip.data_ptr(), occ.data_ptr(), None if count is None else count.data_ptr(),
                                       _stream()), 'march_sweep_x3')
    return occ, skip
