// Alpha composite (transmittance product-scan) forward / backward.
// Reference: stage1/model/rendering.py:196-197 (weights = alpha * cumprod([1, 1-alpha+eps])[:-1]),
// :197 rgb = sum w c, :214-216 acc = sum w, white background; same formula at :405-406.
//
// HBM-bound: forward moves 20*S+16 B/ray, backward 36*S+16 B/ray.  One 64-lane wave per ray.
// Forward: blocked layout (lane l owns the E = ceil(S/64) consecutive samples l*E .. and their 3*E colour floats),
// sequential product inside the lane + one exclusive wave scan over the lane totals, no LDS (see composite_fwd_kernel).
// Backward: strided layout (lane l owns samples l, l+64, ...: coalesced 256 B rows), 6-step wave scans per 64-sample
// chunk with a carried prefix; rgb rows are read as flat coalesced dwords and matched with their sample weight through
// a per-wave LDS row.  Measured (2 M rays x 128 samples): forward 4.84 TB/s, backward 5.09 TB/s (0.61 / 0.64 of the
// 8 TB/s peak; torch's elementwise add reaches 6.1 TB/s on the same box).
#include "common.h"

namespace psn {

constexpr float kEps = 1e-6f;  // rendering.py:8
constexpr int kWavesPerBlock = 4;

// Orders this wave's LDS writes before its later cross-lane LDS reads (LDS ops of one wave execute
// in order; this only has to stop the compiler from moving them).  Waves run different trip counts,
// so a workgroup barrier cannot be used here.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Cross-lane data through DPP (data-parallel primitives: the operand of a VALU instruction is taken from another lane
// of the same row of 16, or broadcast from the row below) instead of __shfl_up (ds_bpermute_b32: an LDS-crossbar round
// trip + a select per step).  Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_from(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143, kWaveShr1 = 0x138;

// inclusive product scan inside every row of 16 lanes (Kogge-Stone, 4 DPP multiplies)
__device__ __forceinline__ float row_incl_prod(float v) {
    v *= dpp_from<kRowShr1, 0xf>(1.0f, v);
    v *= dpp_from<kRowShr2, 0xf>(1.0f, v);
    v *= dpp_from<kRowShr4, 0xf>(1.0f, v);
    v *= dpp_from<kRowShr8, 0xf>(1.0f, v);
    return v;
}
// ... over the whole wave: rows 1 and 3 take lane 15 of the row below, then rows 2 and 3 take lane 31
__device__ __forceinline__ float wave_incl_prod(float v, int /*lane*/) {
    v = row_incl_prod(v);
    v *= dpp_from<kRowBcast15, 0xa>(1.0f, v);
    v *= dpp_from<kRowBcast31, 0xc>(1.0f, v);
    return v;
}
// value of the previous lane of the wave (lane 0: `first`)
__device__ __forceinline__ float wave_prev(float v, float first) { return dpp_from<kWaveShr1, 0xf>(first, v); }
// inclusive sum scan inside a row of 16 lanes: lane 15 of the row ends up with the row total
__device__ __forceinline__ float row_incl_sum(float v) {
    v += dpp_from<kRowShr1, 0xf>(0.0f, v);
    v += dpp_from<kRowShr2, 0xf>(0.0f, v);
    v += dpp_from<kRowShr4, 0xf>(0.0f, v);
    v += dpp_from<kRowShr8, 0xf>(0.0f, v);
    return v;
}
__device__ __forceinline__ float wave_incl_sum_rev(float v, int lane) {  // suffix (inclusive) sum
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v += o;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Forward, blocked layout: lane l owns the E CONSECUTIVE samples l E .. l E + E - 1 of its wave's ray (E = ceil(S / 64)):
// a sequential product inside the lane, ONE exclusive wave scan over the 64 lane totals, no LDS -- a lane already holds
// the colours of its own samples (3 E consecutive floats).  The four ray sums (acc, r, g, b) are reduced together in a
// halving butterfly (7 cross-lane steps instead of 4 x 6).  The previous strided version needed E scans and an LDS
// round trip per ray and was instruction-bound at 0.44 of the HBM peak.  The rows of the NEXT ray of the wave are
// requested before the current one is processed (two register sets, loop unrolled by two).
// streamed once: non-temporal accesses keep the rows out of L2 / Infinity Cache
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 nt_load2(const float* p) {
    const floatx2 v = __builtin_nontemporal_load(reinterpret_cast<const floatx2*>(p));
    return make_float2(v[0], v[1]);
}
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store2(float* p, float a, float b) {
    floatx2 v; v[0] = a; v[1] = b;
    __builtin_nontemporal_store(v, reinterpret_cast<floatx2*>(p));
}
__device__ __forceinline__ void nt_store4(float* p, float a, float b, float c, float d) {
    floatx4 v; v[0] = a; v[1] = b; v[2] = c; v[3] = d;
    __builtin_nontemporal_store(v, reinterpret_cast<floatx4*>(p));
}

template <int E>
struct RayRegs {
    float a[E];
    float c[3 * E];
};

template <int E, bool VEC>
__device__ __forceinline__ void fwd_load(RayRegs<E>& r, const float* __restrict__ alpha, const float* __restrict__ rgb,
                                         int64_t ray, int S, int lane) {
    const int s0 = lane * E;
    const float* a_row = alpha + ray * S + s0;
    const float* c_row = rgb != nullptr ? rgb + (ray * S + s0) * 3 : nullptr;
    if constexpr (VEC) {  // S % E == 0 and 16-byte aligned base: whole lanes are inside or outside the row
        if (s0 < S) {
            if constexpr (E == 2) {
                const float2 t = nt_load2(a_row);
                r.a[0] = t.x; r.a[1] = t.y;
            } else {
#pragma unroll
                for (int i = 0; i < E; i += 4) {
                    const float4 t = nt_load4(a_row + i);
                    r.a[i] = t.x; r.a[i + 1] = t.y; r.a[i + 2] = t.z; r.a[i + 3] = t.w;
                }
            }
            if (c_row != nullptr) {
                if constexpr (E == 2) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float2 t = nt_load2(c_row + 2 * i);
                        r.c[2 * i] = t.x; r.c[2 * i + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 3 * E; i += 4) {
                        const float4 t = nt_load4(c_row + i);
                        r.c[i] = t.x; r.c[i + 1] = t.y; r.c[i + 2] = t.z; r.c[i + 3] = t.w;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) r.a[i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 3 * E; ++i) r.c[i] = 0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) r.a[i] = s0 + i < S ? a_row[i] : 0.0f;
        if (c_row != nullptr) {
#pragma unroll
            for (int i = 0; i < 3 * E; ++i) r.c[i] = s0 + i / 3 < S ? c_row[i] : 0.0f;
        }
    }
}

// Sums of four per-lane values over the wave: lanes 0 / 16 / 32 / 48 end up with the totals of x0 / x1 / x2 / x3.
__device__ __forceinline__ float wave_sum4(float x0, float x1, float x2, float x3, int lane) {
    const bool hi = (lane & 32) != 0;
    const float k0 = (hi ? x2 : x0) + __shfl_xor(hi ? x0 : x2, 32, 64);
    const float k1 = (hi ? x3 : x1) + __shfl_xor(hi ? x1 : x3, 32, 64);
    const bool h16 = (lane & 16) != 0;
    float v = (h16 ? k1 : k0) + __shfl_xor(h16 ? k0 : k1, 16, 64);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <int E, bool VEC>
__device__ __forceinline__ void fwd_ray(const RayRegs<E>& r, bool has_rgb, int64_t ray, int S, int white_bg, int lane,
                                        float* __restrict__ weights, float* __restrict__ rgb_out,
                                        float* __restrict__ acc_out) {
    const int s0 = lane * E;
    float p[E + 1];  // exclusive products inside the lane
    p[0] = 1.0f;
#pragma unroll
    for (int i = 0; i < E; ++i) p[i + 1] = p[i] * (s0 + i < S ? (1.0f - r.a[i] + kEps) : 1.0f);
    const float incl = wave_incl_prod(p[E], lane);
    const float pre = wave_prev(incl, 1.0f);
    float w[E];
    float acc = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        w[i] = r.a[i] * (pre * p[i]);
        acc += w[i];
        q0 += w[i] * r.c[3 * i];
        q1 += w[i] * r.c[3 * i + 1];
        q2 += w[i] * r.c[3 * i + 2];
    }
    if (weights != nullptr) {
        float* w_row = weights + ray * S + s0;
        if constexpr (VEC) {
            if (s0 < S) {
                if constexpr (E == 2) {
                    nt_store2(w_row, w[0], w[1]);
                } else {
#pragma unroll
                    for (int i = 0; i < E; i += 4) nt_store4(w_row + i, w[i], w[i + 1], w[i + 2], w[i + 3]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i)
                if (s0 + i < S) w_row[i] = w[i];
        }
    }
    if (has_rgb) {
        const float v = wave_sum4(acc, q0, q1, q2, lane);
        const float acc_all = __shfl(v, 0, 64);
        if (lane == 0) acc_out[ray] = v;
        if ((lane & 15) == 0 && lane != 0) rgb_out[ray * 3 + (lane >> 4) - 1] = v + (white_bg ? (1.0f - acc_all) : 0.0f);
    } else {
        acc = wave_sum(acc);
        if (lane == 0) acc_out[ray] = acc;
    }
}

template <int E, bool VEC>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ alpha,
                                                            const float* __restrict__ rgb, int64_t n_rays, int S,
                                                            int white_bg, float* __restrict__ weights,
                                                            float* __restrict__ rgb_out, float* __restrict__ acc_out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool has_rgb = rgb != nullptr;
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
    int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (ray >= n_rays) return;
    RayRegs<E> r0, r1;
    fwd_load<E, VEC>(r0, alpha, rgb, ray, S, lane);
    while (true) {
        int64_t nxt = ray + stride;
        if (nxt < n_rays) fwd_load<E, VEC>(r1, alpha, rgb, nxt, S, lane);
        fwd_ray<E, VEC>(r0, has_rgb, ray, S, white_bg, lane, weights, rgb_out, acc_out);
        if (nxt >= n_rays) break;
        ray = nxt;
        nxt = ray + stride;
        if (nxt < n_rays) fwd_load<E, VEC>(r0, alpha, rgb, nxt, S, lane);
        fwd_ray<E, VEC>(r1, has_rgb, ray, S, white_bg, lane, weights, rgb_out, acc_out);
        if (nxt >= n_rays) break;
        ray = nxt;
    }
}

// Forward with every global access a contiguous one over the wave (the backward kernel's layout): lane l owns samples
// l, l + 64, ..., the colours are read as the flat [3 S] row (element k = l + 64 j) and meet their weights through LDS.
// fwd_ray's blocked layout (lane owns E consecutive samples) reads the colours with a 12 E-byte lane stride, which costs
// ~10 % of the achieved HBM rate at S = 128 (4.8-5.1 TB/s there against 5.5 TB/s for the backward kernel).
template <int E>
struct FlatRegs {
    float a[E];
    float c[3 * E];
};
template <int E>
__device__ __forceinline__ void flat_load(FlatRegs<E>& r, const float* __restrict__ alpha, const float* __restrict__ rgb,
                                          int64_t ray, int S, int lane) {
    const float* a_row = alpha + ray * S;
    const float* c_row = rgb + ray * S * 3;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int s = e * 64 + lane;
        r.a[e] = s < S ? __builtin_nontemporal_load(a_row + s) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 3 * E; ++j) {
        const int k = j * 64 + lane;
        r.c[j] = k < 3 * S ? __builtin_nontemporal_load(c_row + k) : 0.0f;
    }
}
template <int E>
__device__ __forceinline__ void flat_ray(const FlatRegs<E>& r, float* __restrict__ wl, int64_t ray, int S, int white_bg,
                                         int lane, float* __restrict__ weights, float* __restrict__ rgb_out,
                                         float* __restrict__ acc_out) {
    float carry = 1.0f, acc = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int s = e * 64 + lane;
        const float t = s < S ? (1.0f - r.a[e] + kEps) : 1.0f;
        const float incl = wave_incl_prod(t, lane);
        const float w = r.a[e] * (wave_prev(incl, 1.0f) * carry);
        if (E > 1) carry *= __shfl(incl, 63, 64);
        acc += w;
        wl[s] = w;
        if (weights != nullptr && s < S) __builtin_nontemporal_store(w, weights + ray * S + s);
    }
    wave_lds_sync();
    // flat element k = lane + 64 j is channel (lane + j) % 3 of sample k / 3: three rotating accumulators
    float q[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3 * E; ++j) {
        const int k = j * 64 + lane;
        q[j % 3] += wl[k < 3 * S ? k / 3 : 0] * r.c[j];
    }
    wave_lds_sync();  // wl is rewritten by the next ray
    const int l3 = lane % 3;
    const float q0 = l3 == 0 ? q[0] : (l3 == 1 ? q[2] : q[1]);
    const float q1 = l3 == 0 ? q[1] : (l3 == 1 ? q[0] : q[2]);
    const float q2 = l3 == 0 ? q[2] : (l3 == 1 ? q[1] : q[0]);
    const float v = wave_sum4(acc, q0, q1, q2, lane);
    const float acc_all = __shfl(v, 0, 64);
    if (lane == 0) acc_out[ray] = v;
    if ((lane & 15) == 0 && lane != 0) rgb_out[ray * 3 + (lane >> 4) - 1] = v + (white_bg ? (1.0f - acc_all) : 0.0f);
}
template <int E>
__global__ __launch_bounds__(256) void composite_fwd_flat_kernel(const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                                 int64_t n_rays, int S, int white_bg, float* __restrict__ weights,
                                                                 float* __restrict__ rgb_out, float* __restrict__ acc_out) {
    __shared__ float lds[kWavesPerBlock][E * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wl = lds[wave];
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
    int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (ray >= n_rays) return;
    FlatRegs<E> r0, r1;
    flat_load<E>(r0, alpha, rgb, ray, S, lane);
    while (true) {
        int64_t nxt = ray + stride;
        if (nxt < n_rays) flat_load<E>(r1, alpha, rgb, nxt, S, lane);
        flat_ray<E>(r0, wl, ray, S, white_bg, lane, weights, rgb_out, acc_out);
        if (nxt >= n_rays) break;
        ray = nxt;
        nxt = ray + stride;
        if (nxt < n_rays) flat_load<E>(r0, alpha, rgb, nxt, S, lane);
        flat_ray<E>(r1, wl, ray, S, white_bg, lane, weights, rgb_out, acc_out);
        if (nxt >= n_rays) break;
        ray = nxt;
    }
}

// Forward with SEVERAL rays per wave (round 6): a ray occupies LPR = 16 (S <= 64) or 32 (S <= 128) lanes, a lane owns four
// consecutive samples -- 16-byte accesses on alpha and the weights (1 KB contiguous per wave instruction over the wave's adjacent
// rays), the transmittance scan is the 4-step DPP row scan (+ one row broadcast for LPR = 32) and the four ray sums are DPP row
// reductions (quad permutes + mirrors: no LDS crossbar).  Measured at 2 M-4 M rays (tools/dbg/experiments/composite_variants.hip,
// profiles/r06b_composite_variants_*.jsonl): S = 64: 5.07 TB/s against 4.7-4.8 for the one-ray-per-wave kernel (whose lanes hold one
// sample each there), S = 96: 5.39 against 4.93; at S = 128 the blocked one-ray kernel stays ahead (4.95-5.1 against 4.7-4.8) and
// keeps the launch.  A plain streaming kernel with the forward's byte mix (4 reads : 1 write, float4 per lane) reaches 4.8-5.2 TB/s
// on the same box: that, not the 6.3 TB/s read-mostly figure, is the ceiling these kernels sit under.
constexpr int kQuad1 = 0xB1, kQuad2 = 0x4E, kHalfMirror = 0x141, kMirror = 0x140;
__device__ __forceinline__ float row_allsum(float v) {
    v += dpp_from<kQuad1, 0xf>(0.0f, v);
    v += dpp_from<kQuad2, 0xf>(0.0f, v);
    v += dpp_from<kHalfMirror, 0xf>(0.0f, v);
    v += dpp_from<kMirror, 0xf>(0.0f, v);
    return v;
}

struct MRegs {
    float a[4];
    float c[12];
};

// MODE 0: temporal loads / stores, 1: non-temporal
template <int LPR, int MODE>
__device__ __forceinline__ void m_load(MRegs& r, const float* __restrict__ alpha, const float* __restrict__ rgb, int64_t ray, int S,
                                       int sub, bool live) {
    const int s0 = sub * 4;
    if (live && s0 < S) {
        const float* ap = alpha + ray * S + s0;
        const float* cp = rgb + (ray * S + s0) * 3;
        float4 t = MODE == 1 ? nt_load4(ap) : *reinterpret_cast<const float4*>(ap);
        r.a[0] = t.x; r.a[1] = t.y; r.a[2] = t.z; r.a[3] = t.w;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 u = MODE == 1 ? nt_load4(cp + 4 * i) : *reinterpret_cast<const float4*>(cp + 4 * i);
            r.c[4 * i] = u.x; r.c[4 * i + 1] = u.y; r.c[4 * i + 2] = u.z; r.c[4 * i + 3] = u.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.a[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 12; ++i) r.c[i] = 0.0f;
    }
}

template <int LPR, int MODE>
__device__ __forceinline__ void m_ray(const MRegs& r, int64_t ray, int S, int white_bg, int lane, int sub, bool live,
                                      float* __restrict__ weights, float* __restrict__ rgb_out, float* __restrict__ acc_out) {
    const int s0 = sub * 4;
    const bool in = s0 < S;
    float p[5];
    p[0] = 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i + 1] = p[i] * (in ? (1.0f - r.a[i] + kEps) : 1.0f);
    float incl = row_incl_prod(p[4]);
    float pre;
    if constexpr (LPR == 32) {
        incl *= dpp_from<kRowBcast15, 0xa>(1.0f, incl);
        pre = wave_prev(incl, 1.0f);
        if ((lane & 31) == 0) pre = 1.0f;
    } else {
        pre = dpp_from<kRowShr1, 0xf>(1.0f, incl);
    }
    float w[4];
    float acc = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[i] = r.a[i] * (pre * p[i]);
        acc += w[i];
        q0 += w[i] * r.c[3 * i];
        q1 += w[i] * r.c[3 * i + 1];
        q2 += w[i] * r.c[3 * i + 2];
    }
    if (weights != nullptr && live && in) {
        float* wp = weights + ray * S + s0;
        if (MODE == 1) nt_store4(wp, w[0], w[1], w[2], w[3]);
        else *reinterpret_cast<float4*>(wp) = make_float4(w[0], w[1], w[2], w[3]);
    }
    acc = row_allsum(acc); q0 = row_allsum(q0); q1 = row_allsum(q1); q2 = row_allsum(q2);
    if constexpr (LPR == 32) {
        acc += dpp_from<kRowBcast15, 0xa>(0.0f, acc);
        q0 += dpp_from<kRowBcast15, 0xa>(0.0f, q0);
        q1 += dpp_from<kRowBcast15, 0xa>(0.0f, q1);
        q2 += dpp_from<kRowBcast15, 0xa>(0.0f, q2);
    }
    if (live && sub == LPR - 1) {
        const float bg = white_bg ? (1.0f - acc) : 0.0f;
        acc_out[ray] = acc;
        rgb_out[ray * 3 + 0] = q0 + bg;
        rgb_out[ray * 3 + 1] = q1 + bg;
        rgb_out[ray * 3 + 2] = q2 + bg;
    }
}

template <int LPR, int MODE, int DEPTH>
__global__ __launch_bounds__(256) void composite_fwd_multi_kernel(const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                                  int64_t n_rays, int S, int white_bg, float* __restrict__ weights,
                                                                  float* __restrict__ rgb_out, float* __restrict__ acc_out) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), rq = lane / LPR;
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * RPW;
    int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * RPW;
    if (base >= n_rays) return;
    if constexpr (DEPTH == 1) {
        MRegs r0, r1;
        m_load<LPR, MODE>(r0, alpha, rgb, base + rq, S, sub, base + rq < n_rays);
        while (true) {
            int64_t nxt = base + stride;
            if (nxt < n_rays) m_load<LPR, MODE>(r1, alpha, rgb, nxt + rq, S, sub, nxt + rq < n_rays);
            m_ray<LPR, MODE>(r0, base + rq, S, white_bg, lane, sub, base + rq < n_rays, weights, rgb_out, acc_out);
            if (nxt >= n_rays) break;
            base = nxt;
            nxt = base + stride;
            if (nxt < n_rays) m_load<LPR, MODE>(r0, alpha, rgb, nxt + rq, S, sub, nxt + rq < n_rays);
            m_ray<LPR, MODE>(r1, base + rq, S, white_bg, lane, sub, base + rq < n_rays, weights, rgb_out, acc_out);
            if (nxt >= n_rays) break;
            base = nxt;
        }
    } else {  // two ray sets ahead (three register sets)
        MRegs r0, r1, r2;
        m_load<LPR, MODE>(r0, alpha, rgb, base + rq, S, sub, base + rq < n_rays);
        if (base + stride < n_rays) m_load<LPR, MODE>(r1, alpha, rgb, base + stride + rq, S, sub, base + stride + rq < n_rays);
        while (true) {
            int64_t n2 = base + 2 * stride;
            if (n2 < n_rays) m_load<LPR, MODE>(r2, alpha, rgb, n2 + rq, S, sub, n2 + rq < n_rays);
            m_ray<LPR, MODE>(r0, base + rq, S, white_bg, lane, sub, base + rq < n_rays, weights, rgb_out, acc_out);
            base += stride; if (base >= n_rays) break;
            n2 = base + 2 * stride;
            if (n2 < n_rays) m_load<LPR, MODE>(r0, alpha, rgb, n2 + rq, S, sub, n2 + rq < n_rays);
            m_ray<LPR, MODE>(r1, base + rq, S, white_bg, lane, sub, base + rq < n_rays, weights, rgb_out, acc_out);
            base += stride; if (base >= n_rays) break;
            n2 = base + 2 * stride;
            if (n2 < n_rays) m_load<LPR, MODE>(r1, alpha, rgb, n2 + rq, S, sub, n2 + rq < n_rays);
            m_ray<LPR, MODE>(r2, base + rq, S, white_bg, lane, sub, base + rq < n_rays, weights, rgb_out, acc_out);
            base += stride; if (base >= n_rays) break;
        }
    }
}

// Accumulated opacity only (shadow rays, stage1/model/rendering.py:405-406: no colours, no weights kept), S <= 128 and a
// multiple of 16: FOUR rays per wave -- a ray occupies one DPP row of 16 lanes, a lane owns E = S / 16 consecutive samples
// (16-byte loads), the transmittance scan is the 4-step row scan and the ray sum a 4-step row reduction.  With one ray
// per wave (E = 2 at S = 128) the cross-lane steps outnumbered the loads: 2.6 TB/s.
template <int E>
__global__ __launch_bounds__(256) void composite_acc4_kernel(const float* __restrict__ alpha, int64_t n_rays, int S,
                                                             float* __restrict__ acc_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rq = lane >> 4;
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 4;
    for (int64_t ray = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 4 + rq; ray < n_rays + 3; ray += stride) {
        const bool live = ray < n_rays;  // (the loop bound keeps whole waves together for the DPP steps)
        float a[E];
        const int s0 = sub * E;
        if (live && s0 < S) {
#pragma unroll
            for (int i = 0; i < E; i += 4) {
                const float4 t = nt_load4(alpha + ray * S + s0 + i);
                a[i] = t.x; a[i + 1] = t.y; a[i + 2] = t.z; a[i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) a[i] = 0.0f;
        }
        float p[E + 1];
        p[0] = 1.0f;
#pragma unroll
        for (int i = 0; i < E; ++i) p[i + 1] = p[i] * (s0 + i < S ? (1.0f - a[i] + kEps) : 1.0f);
        const float incl = row_incl_prod(p[E]);
        const float pre = dpp_from<kRowShr1, 0xf>(1.0f, incl);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < E; ++i) acc += a[i] * (pre * p[i]);
        acc = row_incl_sum(acc);
        if (live && sub == 15) acc_out[ray] = acc;
    }
}

template <int E>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ alpha,
                                                            const float* __restrict__ rgb,
                                                            const float* __restrict__ d_rgb_out,
                                                            const float* __restrict__ d_acc_out, int64_t n_rays,
                                                            int S, int white_bg, float* __restrict__ d_alpha,
                                                            float* __restrict__ d_rgb) {
    // LDS per wave: w[S] then gc[3S] (flat g_c * (c - wb))
    __shared__ float lds[kWavesPerBlock][E * 64 * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* wl = lds[wave];
    float* gl = wl + E * 64;
    const float wb = white_bg ? 1.0f : 0.0f;
    for (int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave; ray < n_rays;
         ray += (int64_t)gridDim.x * kWavesPerBlock) {
        const float* a_row = alpha + ray * S;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (rgb != nullptr) {
            g0 = d_rgb_out[ray * 3 + 0];
            g1 = d_rgb_out[ray * 3 + 1];
            g2 = d_rgb_out[ray * 3 + 2];
        }
        const float gacc = d_acc_out != nullptr ? d_acc_out[ray] : 0.0f;
        float a[E], T[E], w[E];
        float carry = 1.0f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int s = e * 64 + lane;
            a[e] = s < S ? __builtin_nontemporal_load(a_row + s) : 0.0f;
            float t = s < S ? (1.0f - a[e] + kEps) : 1.0f;
            float incl = wave_incl_prod(t, lane);
            float excl = wave_prev(incl, 1.0f);
            T[e] = excl * carry;
            carry *= __shfl(incl, 63, 64);
            w[e] = a[e] * T[e];
            wl[e * 64 + lane] = w[e];
        }
        wave_lds_sync();
        const int n_flat = 3 * S;
        if (rgb != nullptr) {
            const float* c_row = rgb + ray * S * 3;
            float* dc_row = d_rgb + ray * S * 3;
            for (int k = lane; k < n_flat; k += 64) {
                float c = __builtin_nontemporal_load(c_row + k);
                int s = k / 3;
                int ch = k - 3 * s;
                float g = ch == 0 ? g0 : (ch == 1 ? g1 : g2);
                gl[k] = g * (c - wb);
                __builtin_nontemporal_store(wl[s] * g, dc_row + k);
            }
            wave_lds_sync();
        }
        // G_s = dL/dw_s ; suffix sums R_s = sum_{k>s} G_k w_k (processed from the last chunk backwards)
        float tail = 0.0f;  // sum over later chunks
#pragma unroll
        for (int e = E - 1; e >= 0; --e) {
            int s = e * 64 + lane;
            float G = gacc;
            if (rgb != nullptr && s < S) G += gl[3 * s] + gl[3 * s + 1] + gl[3 * s + 2];
            float gw = s < S ? G * w[e] : 0.0f;
            // exclusive suffix sum computed directly (incl - self would cancel catastrophically when the
            // own term dominates, e.g. alpha == 1 rows where the result is then divided by t ~ 1e-6)
            float nxt = __shfl_down(gw, 1, 64);
            if (lane == 63) nxt = 0.0f;
            float ex = wave_incl_sum_rev(nxt, lane);
            float R = ex + tail;
            tail += __shfl(ex, 0, 64) + __shfl(gw, 0, 64);
            if (s < S) {
                float t = 1.0f - a[e] + kEps;
                __builtin_nontemporal_store(G * T[e] - R / t, d_alpha + ray * S + s);
            }
        }
    }
}

template <int E>
static int launch_fwd(const float* alpha, const float* rgb, int64_t n_rays, int S, int white_bg, float* weights,
                      float* rgb_out, float* acc_out, hipStream_t st) {
    int64_t blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // vector path: every lane's E samples lie entirely inside or outside the row and all of its accesses are aligned
    // (E = 2: 8-byte accesses at even sample offsets; E >= 4: 16-byte accesses at multiples of four samples)
    const bool vec = E >= 2 && S % E == 0 && (((uintptr_t)alpha | (uintptr_t)rgb | (uintptr_t)weights) & 15) == 0;
    if (vec) hipLaunchKernelGGL((composite_fwd_kernel<E, true>), dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, S, white_bg, weights, rgb_out, acc_out);
    else hipLaunchKernelGGL((composite_fwd_kernel<E, false>), dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, S, white_bg, weights, rgb_out, acc_out);
    PSN_CHECK_LAUNCH("composite_fwd");
    return PSN_OK;
}
template <int E>
static int launch_bwd(const float* alpha, const float* rgb, const float* d_rgb_out, const float* d_acc_out,
                      int64_t n_rays, int S, int white_bg, float* d_alpha, float* d_rgb, hipStream_t st) {
    int64_t blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(composite_bwd_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, d_rgb_out,
                       d_acc_out, n_rays, S, white_bg, d_alpha, d_rgb);
    PSN_CHECK_LAUNCH("composite_bwd");
    return PSN_OK;
}

}  // namespace psn

extern "C" int psn_composite_fwd(const float* alpha, const float* rgb, int64_t n_rays, int n_samples, int white_bg,
                                 float* weights, float* rgb_out, float* acc_out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(alpha && acc_out, "composite_fwd: null alpha/acc_out");
    PSN_CHECK_ARG((rgb == nullptr) == (rgb_out == nullptr), "composite_fwd: rgb and rgb_out must both be set or both null");
    PSN_CHECK_ARG(n_samples >= 1 && n_samples <= 1024, "composite_fwd: n_samples=%d out of [1,1024]", n_samples);
    if (n_rays <= 0) return PSN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (rgb == nullptr && weights == nullptr && n_samples % 16 == 0 && n_samples / 16 >= 4 && n_samples <= 128 &&
        (n_samples / 16) % 4 == 0 && ((uintptr_t)alpha & 15) == 0) {  // opacity only: four rays per wave
        int64_t blocks = (n_rays + kWavesPerBlock * 4 - 1) / (kWavesPerBlock * 4);
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (n_samples == 64) hipLaunchKernelGGL(composite_acc4_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, n_rays, n_samples, acc_out);
        else hipLaunchKernelGGL(composite_acc4_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, n_rays, n_samples, acc_out);
        PSN_CHECK_LAUNCH("composite_fwd (opacity only)");
        return PSN_OK;
    }
    // measured at S = 128, 2 M rays (tools/bench_composite.py, same box, 3 runs): without the weights output the flat layout
    // reaches 5.32-5.34 TB/s against 4.69-4.72 for the blocked one; with it the blocked one keeps a 2 % lead (4.93 vs 4.84)
    if (weights == nullptr && rgb != nullptr && n_samples > 64 && n_samples <= 128) {
        int64_t blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(composite_fwd_flat_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, n_samples,
                           white_bg, weights, rgb_out, acc_out);
        PSN_CHECK_LAUNCH("composite_fwd (flat layout)");
        return PSN_OK;
    }
    if (rgb != nullptr && weights != nullptr && n_samples % 4 == 0 && n_samples < 128 &&
        (((uintptr_t)alpha | (uintptr_t)rgb | (uintptr_t)weights) & 15) == 0) {
        // several rays per wave, four samples per lane (S <= 64: 4 rays, non-temporal accesses; 64 < S < 128: 2 rays, temporal)
        const int rpw = n_samples <= 64 ? 4 : 2;
        int64_t blocks = (n_rays + kWavesPerBlock * rpw - 1) / (kWavesPerBlock * rpw);
        if (n_samples <= 64) {
            if (blocks > 256 * 8) blocks = 256 * 8;
            hipLaunchKernelGGL((composite_fwd_multi_kernel<16, 1, 1>), dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out);
        } else {
            if (blocks > 256 * 16) blocks = 256 * 16;
            hipLaunchKernelGGL((composite_fwd_multi_kernel<32, 0, 1>), dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out);
        }
        PSN_CHECK_LAUNCH("composite_fwd (several rays per wave)");
        return PSN_OK;
    }
    int E = (n_samples + 63) / 64;
    if (E <= 1) return launch_fwd<1>(alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out, st);
    if (E <= 2) return launch_fwd<2>(alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out, st);
    if (E <= 4) return launch_fwd<4>(alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out, st);
    if (E <= 8) return launch_fwd<8>(alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out, st);
    return launch_fwd<16>(alpha, rgb, n_rays, n_samples, white_bg, weights, rgb_out, acc_out, st);
}

extern "C" int psn_composite_bwd(const float* alpha, const float* rgb, const float* d_rgb_out, const float* d_acc_out,
                                 int64_t n_rays, int n_samples, int white_bg, float* d_alpha, float* d_rgb,
                                 void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(alpha && d_alpha, "composite_bwd: null alpha/d_alpha");
    PSN_CHECK_ARG((rgb == nullptr) == (d_rgb == nullptr) && (rgb == nullptr) == (d_rgb_out == nullptr),
                  "composite_bwd: rgb, d_rgb_out, d_rgb must all be set or all null");
    PSN_CHECK_ARG(n_samples >= 1 && n_samples <= 1024, "composite_bwd: n_samples=%d out of [1,1024]", n_samples);
    if (n_rays <= 0) return PSN_OK;
    hipStream_t st = (hipStream_t)stream;
    int E = (n_samples + 63) / 64;
    if (E <= 1) return launch_bwd<1>(alpha, rgb, d_rgb_out, d_acc_out, n_rays, n_samples, white_bg, d_alpha, d_rgb, st);
    if (E <= 2) return launch_bwd<2>(alpha, rgb, d_rgb_out, d_acc_out, n_rays, n_samples, white_bg, d_alpha, d_rgb, st);
    if (E <= 4) return launch_bwd<4>(alpha, rgb, d_rgb_out, d_acc_out, n_rays, n_samples, white_bg, d_alpha, d_rgb, st);
    if (E <= 8) return launch_bwd<8>(alpha, rgb, d_rgb_out, d_acc_out, n_rays, n_samples, white_bg, d_alpha, d_rgb, st);
    return launch_bwd<16>(alpha, rgb, d_rgb_out, d_acc_out, n_rays, n_samples, white_bg, d_alpha, d_rgb, st);
}
