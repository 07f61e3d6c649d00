// Positional encodings: stage1 PositionalEncoding (stage1/model/network.py:141-150, pi factor 1.0) and
// stage2 Embedder (stage2/model/embedder.py:6-54, log-sampled 2^k bands).  Both produce
//   [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...].
// 2^k scaling is exact in fp32, so the only rounding is sinf/cosf (full-range ocml versions).
#include "common.h"

namespace psn {

__global__ __launch_bounds__(256) void pe_encode_kernel(const float* __restrict__ x, int64_t n, int n_freqs,
                                                        float scale, float* __restrict__ out, int out_stride) {
    const int64_t total = n * out_stride;
    const int width = 3 + 6 * n_freqs;
    const bool pow64 = out_stride == 64;  // the usual table width: shift / mask instead of a 64-bit division per element
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t row = pow64 ? (e >> 6) : e / out_stride;
        int col = (int)(e - row * out_stride);
        float v = 0.0f;
        if (col < 3) {
            v = x[row * 3 + col] * scale;
        } else if (col < width) {
            int q = col - 3;
            int f = q / 6, w = q - 6 * f;
            int c = w % 3;
            float arg = ldexpf(x[row * 3 + c] * scale, f);
            v = (w >= 3) ? cosf(arg) : sinf(arg);
        }
        out[e] = v;
    }
}

__global__ __launch_bounds__(256) void pe_encode_jvp_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            int64_t n, int n_freqs, float scale,
                                                            float* __restrict__ out, int out_stride) {
    const int64_t total = n * out_stride;
    const int width = 3 + 6 * n_freqs;
    const bool pow64 = out_stride == 64;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t row = pow64 ? (e >> 6) : e / out_stride;
        int col = (int)(e - row * out_stride);
        float v = 0.0f;
        if (col < 3) {
            v = t[row * 3 + col] * scale;
        } else if (col < width) {
            int q = col - 3;
            int f = q / 6, w = q - 6 * f;
            int c = w % 3;
            float arg = ldexpf(x[row * 3 + c] * scale, f);
            float d = (w >= 3) ? -sinf(arg) : cosf(arg);
            v = ldexpf(d * (t[row * 3 + c] * scale), f);
        }
        out[e] = v;
    }
}

// The 64-column table forms of the two kernels above (the width every fused engine reads): one thread per four consecutive
// columns -- the point is loaded once per thread, sincosf shares the range reduction of the band's sin and cos (the same bits
// as sinf / cosf: tests compare the table path with the in-kernel encoding of mlp_infer.hip bit for bit), 16-byte stores.
template <bool JVP>
__global__ __launch_bounds__(256) void pe_encode64_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                          int n_freqs, float scale, float* __restrict__ out) {
    const int64_t total = n * 16;
    const int width = 3 + 6 * n_freqs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e >> 4;
        const int c0 = (int)(e & 15) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < width) {
            const float x0 = x[row * 3 + 0] * scale, x1 = x[row * 3 + 1] * scale, x2 = x[row * 3 + 2] * scale;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
            if constexpr (JVP) { t0 = t[row * 3 + 0] * scale; t1 = t[row * 3 + 1] * scale; t2 = t[row * 3 + 2] * scale; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = c0 + i;
                if (col < 3) {
                    o[i] = JVP ? (col == 0 ? t0 : (col == 1 ? t1 : t2)) : (col == 0 ? x0 : (col == 1 ? x1 : x2));
                } else if (col < width) {
                    const int q = col - 3;
                    const int f = q / 6, w = q - 6 * f;
                    const int c = w % 3;
                    float sn, cs;
                    sincosf(ldexpf(c == 0 ? x0 : (c == 1 ? x1 : x2), f), &sn, &cs);
                    if constexpr (JVP) {
                        const float d = (w >= 3) ? -sn : cs;
                        o[i] = ldexpf(d * (c == 0 ? t0 : (c == 1 ? t1 : t2)), f);
                    } else {
                        o[i] = (w >= 3) ? cs : sn;
                    }
                }
            }
        }
        *reinterpret_cast<float4*>(out + row * 64 + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// d_out2 (or nullptr): a second gradient of the encoding that is ADDED column by column before the chain rule (the sweep
// of the stage-1 geometry network delivers d logit / d pe in two pieces: the layer-0 columns of one dump and the skip
// layer's columns of another; reading both here saves a [Q, 64] copy and a [Q, 39] add per call)
__global__ __launch_bounds__(256) void pe_encode_bwd_kernel(const float* __restrict__ x, const float* __restrict__ d_out,
                                                            int64_t n, int n_freqs, float scale, int out_stride,
                                                            const float* __restrict__ d_out2, int out2_stride,
                                                            float* __restrict__ d_x) {
    const int64_t total = n * 3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t row = e / 3;
        int c = (int)(e - row * 3);
        const float* g = d_out + row * out_stride;
        const float* g2 = d_out2 != nullptr ? d_out2 + row * out2_stride : nullptr;
        float xs = x[e] * scale;
        float acc = g2 != nullptr ? g[c] + g2[c] : g[c];
        for (int f = 0; f < n_freqs; ++f) {
            float arg = ldexpf(xs, f);
            float s, co;
            sincosf(arg, &s, &co);
            float gs = g[3 + 6 * f + c], gc = g[3 + 6 * f + 3 + c];
            if (g2 != nullptr) { gs += g2[3 + 6 * f + c]; gc += g2[3 + 6 * f + 3 + c]; }
            acc += ldexpf(co * gs - s * gc, f);
        }
        d_x[e] = acc * scale;
    }
}

}  // namespace psn

extern "C" int psn_pe_encode(const float* x, int64_t n, int n_freqs, float scale, float* out, int out_stride,
                             void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && out, "pe_encode: null pointer");
    PSN_CHECK_ARG(n_freqs >= 0 && n_freqs <= 16 && out_stride >= 3 + 6 * n_freqs, "pe_encode: n_freqs=%d out_stride=%d", n_freqs, out_stride);
    if (n <= 0) return PSN_OK;
    int64_t total = n * out_stride;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (out_stride == 64 && (((uintptr_t)out) & 15) == 0) {
        int64_t b4 = (n * 16 + 255) / 256;
        if (b4 > 256 * 32) b4 = 256 * 32;
        hipLaunchKernelGGL(pe_encode64_kernel<false>, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, x, nullptr, n, n_freqs, scale, out);
    } else {
        hipLaunchKernelGGL(pe_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, n_freqs, scale, out, out_stride);
    }
    PSN_CHECK_LAUNCH("pe_encode");
    return PSN_OK;
}

namespace psn {
// Input table of the stage-1 appearance network (stage1/model/network.py:128-138: cat[p, gamma(v / |v|), n, features] -- the
// features enter the fused chain as initial activations, the rest as this 64-column table): row r = [p (3) | v^ (3) |
// sin / cos bands of v^ (6 n_freqs) | n (3) | 0 ...] with v^ = v / |v|, in ONE launch instead of a zero fill, three strided
// copies, the norm / division kernels and the encoding.  Same expressions as pe_encode_kernel for the bands.
__global__ __launch_bounds__(256) void app_input_kernel(const float* __restrict__ p, const float* __restrict__ v,
                                                        const float* __restrict__ nrm, int64_t n, int n_freqs,
                                                        float* __restrict__ out) {
    const int64_t total = n * 16;  // one thread per four consecutive columns (one 16-byte store)
    const int dv = 3 + 6 * n_freqs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e >> 4;
        const int c0 = (int)(e & 15) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < 6 + dv) {
            float vx = 0.f, vy = 0.f, vz = 0.f, len = 1.f;
            if (c0 + 3 >= 3 && c0 < 3 + dv) {
                vx = v[row * 3 + 0]; vy = v[row * 3 + 1]; vz = v[row * 3 + 2];
                len = sqrtf(vx * vx + vy * vy + vz * vz);  // torch.norm(v, dim=-1)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = c0 + i;
                if (col < 3) {
                    o[i] = p[row * 3 + col];
                } else if (col < 3 + dv) {
                    const int q = col - 3;
                    if (q < 3) {
                        o[i] = (q == 0 ? vx : (q == 1 ? vy : vz)) / len;
                    } else {
                        const int f = (q - 3) / 6, w = (q - 3) - 6 * f;
                        const int c = w % 3;
                        float sn, cs;  // one range reduction for both: the same bits as sinf / cosf (mlp_infer.hip, SRC == 2)
                        sincosf(ldexpf((c == 0 ? vx : (c == 1 ? vy : vz)) / len, f), &sn, &cs);
                        o[i] = (w >= 3) ? cs : sn;
                    }
                } else if (col < 6 + dv) {
                    o[i] = nrm[row * 3 + (col - 3 - dv)];
                }
            }
        }
        *reinterpret_cast<float4*>(out + row * 64 + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace psn

extern "C" int psn_app_input(const float* p, const float* v, const float* normal, int64_t n, int n_freqs, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(p && v && normal && out, "app_input: null pointer");
    PSN_CHECK_ARG(n_freqs >= 0 && 9 + 6 * n_freqs <= 64, "app_input: n_freqs=%d (3 + (3 + 6 n_freqs) + 3 columns must fit 64)", n_freqs);
    if (n <= 0) return PSN_OK;
    int64_t blocks = (n * 16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(app_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, v, normal, n, n_freqs, out);
    PSN_CHECK_LAUNCH("app_input");
    return PSN_OK;
}

extern "C" int psn_pe_encode_jvp(const float* x, const float* t, int64_t n, int n_freqs, float scale, float* out,
                                 int out_stride, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && t && out, "pe_encode_jvp: null pointer");
    PSN_CHECK_ARG(n_freqs >= 0 && n_freqs <= 16 && out_stride >= 3 + 6 * n_freqs, "pe_encode_jvp: n_freqs=%d out_stride=%d", n_freqs, out_stride);
    if (n <= 0) return PSN_OK;
    int64_t total = n * out_stride;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (out_stride == 64 && (((uintptr_t)out) & 15) == 0) {
        int64_t b4 = (n * 16 + 255) / 256;
        if (b4 > 256 * 32) b4 = 256 * 32;
        hipLaunchKernelGGL(pe_encode64_kernel<true>, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, x, t, n, n_freqs, scale, out);
    } else {
        hipLaunchKernelGGL(pe_encode_jvp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, t, n, n_freqs, scale, out, out_stride);
    }
    PSN_CHECK_LAUNCH("pe_encode_jvp");
    return PSN_OK;
}

extern "C" int psn_pe_encode_bwd(const float* x, const float* d_out, int64_t n, int n_freqs, float scale, int out_stride,
                                 const float* d_out2, int out2_stride, float* d_x, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && d_out && d_x, "pe_encode_bwd: null pointer");
    PSN_CHECK_ARG(d_out2 == nullptr || out2_stride >= 3 + 6 * n_freqs, "pe_encode_bwd: out2_stride=%d", out2_stride);
    PSN_CHECK_ARG(n_freqs >= 0 && n_freqs <= 16 && out_stride >= 3 + 6 * n_freqs, "pe_encode_bwd: n_freqs=%d out_stride=%d", n_freqs, out_stride);
    if (n <= 0) return PSN_OK;
    int64_t blocks = (n * 3 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pe_encode_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, d_out, n, n_freqs, scale, out_stride, d_out2,
                       out2_stride, d_x);
    PSN_CHECK_LAUNCH("pe_encode_bwd");
    return PSN_OK;
}

// ---- dense outputs of the stage-2 model (stage2/model/renderer.py:145-152, 204-264) -----------------------------------
// The reference returns per-pixel tensors [B, N, C] pre-filled with a constant (1, or 0 for the SG weights) and carrying
// the surface rows at the pixels of the surface mask.  All of them (up to PSN_SCATTER_MAX_ITEMS per call) are written by
// ONE launch instead of a fill + an index_copy each; `inv` [N] maps a pixel to its surface row or -1.  HBM-bound.
namespace psn {
struct ScatterArgs {
    PsnScatterItem it[PSN_SCATTER_MAX_ITEMS];
    int64_t start[PSN_SCATTER_MAX_ITEMS + 1];  // first 256-element chunk of each item
    const int* inv;
    const int64_t* idx;
    int64_t N, Ns;
    int n;
};
// One thread per (b, pixel) -- not per element: the index arithmetic (one 32-bit division) is paid once per pixel and
// the C values of a pixel are written as one contiguous run.  (The per-element form spent its time in three 64-bit
// integer divisions per float: 109 us for the 113 MB of a bear.conf step; this one is HBM-bound.)
__global__ __launch_bounds__(256) void scatter_rows_kernel(ScatterArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int64_t)blockIdx.x >= a.start[i + 1]) ++i;
    const PsnScatterItem it = a.it[i];
    const uint32_t bn = (uint32_t)(((int64_t)blockIdx.x - a.start[i]) * 256 + threadIdx.x);  // (b, pixel): B * N < 2^31 (host check)
    const uint32_t N = (uint32_t)a.N;
    if (bn >= (uint32_t)it.B * N) return;
    const uint32_t b = bn / N, n = bn - b * N;
    const int r = a.inv[n];
    float* dst = it.dense + (int64_t)bn * it.C;
    if (r >= 0) {
        const float* src = it.rows + ((int64_t)b * a.Ns + r) * it.row_stride;
        for (int c = 0; c < it.C; ++c) dst[c] = src[(int64_t)c * it.col_stride];
    } else {
        for (int c = 0; c < it.C; ++c) dst[c] = it.fill;
    }
}
// adjoint: rows_grad[(b Ns + r), c] = dense_grad[b, idx[r], c]; one thread per element (coalesced writes), 32-bit index
// arithmetic (B * Ns * C < 2^31, host check)
__global__ __launch_bounds__(256) void gather_rows_kernel(ScatterArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int64_t)blockIdx.x >= a.start[i + 1]) ++i;
    const PsnScatterItem it = a.it[i];
    const uint32_t e = (uint32_t)(((int64_t)blockIdx.x - a.start[i]) * 256 + threadIdx.x);  // element of rows_grad [B Ns, C]
    const uint32_t Ns = (uint32_t)a.Ns, C = (uint32_t)it.C;
    if (e >= (uint32_t)it.B * Ns * C) return;
    const uint32_t br = e / C, c = e - br * C;
    const uint32_t b = br / Ns, r = br - b * Ns;
    const int64_t px = a.idx[r];
    // a PADDED index list (psn_surface_index: the last surface pixel repeated up to a fixed capacity) maps its pixel back to
    // the FIRST of the equal entries: every later one is a dead row and gets an exact zero, not a second copy of the gradient
    const bool live = a.inv == nullptr || a.inv[px] == (int)r;
    const_cast<float*>(it.rows)[e] = live ? it.dense[((int64_t)b * a.N + px) * it.C + c] : 0.0f;
}

static int launch_scatter(int n_items, const PsnScatterItem* items, const int* inv, const int64_t* idx, int64_t N, int64_t Ns,
                          bool gather, void* stream) {
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= PSN_SCATTER_MAX_ITEMS, "scatter_rows: n_items=%d", n_items);
    PSN_CHECK_ARG(gather ? idx != nullptr : inv != nullptr, "scatter_rows: missing index map");
    PSN_CHECK_ARG(N >= 1 && Ns >= 0, "scatter_rows: bad sizes");
    ScatterArgs a;
    a.n = n_items; a.inv = inv; a.idx = idx; a.N = N; a.Ns = Ns;
    a.start[0] = 0;
    for (int i = 0; i < n_items; ++i) {
        const PsnScatterItem& it = items[i];
        PSN_CHECK_ARG(it.dense && (it.rows || Ns == 0) && it.B >= 1 && it.C >= 1, "scatter_rows: item %d: null pointer or empty shape", i);
        a.it[i] = it;
        const int64_t threads = gather ? (int64_t)it.B * Ns * it.C : (int64_t)it.B * N;  // per element / per (b, pixel)
        PSN_CHECK_ARG(threads < (1ll << 31), "scatter_rows: item %d: B * N (scatter) / B * Ns * C (gather) must stay below 2^31", i);
        a.start[i + 1] = a.start[i] + (threads + 255) / 256;
    }
    if (a.start[n_items] == 0) return PSN_OK;
    PSN_CHECK_ARG(a.start[n_items] < (1ll << 31), "scatter_rows: too many elements");
    if (gather) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)a.start[n_items]), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)a.start[n_items]), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH(gather ? "gather_rows" : "scatter_rows");
    return PSN_OK;
}
}  // namespace psn

extern "C" int psn_scatter_rows(int n_items, const PsnScatterItem* items, const int* inv, int64_t N, int64_t Ns, void* stream) {
    return psn::launch_scatter(n_items, items, inv, nullptr, N, Ns, false, stream);
}
extern "C" int psn_gather_rows_valid(int n_items, const PsnScatterItem* items, const int64_t* idx, const int* inv, int64_t N, int64_t Ns,
                                     void* stream) {
    return psn::launch_scatter(n_items, items, inv, idx, N, Ns, true, stream);
}
extern "C" int psn_gather_rows(int n_items, const PsnScatterItem* items, const int64_t* idx, int64_t N, int64_t Ns, void* stream) {
    return psn::launch_scatter(n_items, items, nullptr, idx, N, Ns, true, stream);
}
