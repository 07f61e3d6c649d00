// Alpha composite (transmittance product-scan) forward / backward.
// Reference: stage1/model/rendering.py:196-197 (weights = alpha * cumprod([1, 1-alpha+eps])[:-1]),
// :197 rgb = sum w c, :214-216 acc = sum w, white background; same formula at :405-406.
//
// HBM-bound: forward moves 20*S+16 B/ray, backward 36*S+16 B/ray.
// One 64-lane wave per ray: lane l owns samples l, l+64, ... (coalesced 256 B rows); the exclusive
// product scan runs as a 6-step wave scan per 64-sample chunk with a carried prefix; rgb rows (3*S
// floats, not float4-aligned per sample) are read as flat coalesced dwords and matched with their
// sample weight through a per-wave LDS row.
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

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d, 64);
        if (lane >= d) v *= o;
    }
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

template <int E>  // E = chunks of 64 samples held per lane
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ alpha,
                                                            const float* __restrict__ rgb, int64_t n_rays, int S,
                                                            int white_bg, float* __restrict__ weights,
                                                            float* __restrict__ rgb_out, float* __restrict__ acc_out) {
    __shared__ float w_lds[kWavesPerBlock][E * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* wl = w_lds[wave];
    for (int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave; ray < n_rays;
         ray += (int64_t)gridDim.x * kWavesPerBlock) {
        const float* a_row = alpha + ray * S;
        float a[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int s = e * 64 + lane;
            a[e] = s < S ? a_row[s] : 0.0f;
        }
        float carry = 1.0f, acc = 0.0f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int s = e * 64 + lane;
            float t = s < S ? (1.0f - a[e] + kEps) : 1.0f;
            float incl = wave_incl_prod(t, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            float w = a[e] * (excl * carry);
            carry *= __shfl(incl, 63, 64);
            acc += w;
            wl[e * 64 + lane] = w;
            if (weights != nullptr && s < S) weights[ray * S + s] = w;
        }
        acc = wave_sum(acc);
        if (rgb != nullptr) {
            // wave-private LDS row: writes above are visible to this wave's reads after the waitcnt
            wave_lds_sync();
            const float* c_row = rgb + ray * S * 3;
            float part0 = 0.f, part1 = 0.f, part2 = 0.f;
            const int n_flat = 3 * S;
            for (int k = lane; k < n_flat; k += 64) {
                float c = c_row[k];
                int s = k / 3;
                int ch = k - 3 * s;
                float p = wl[s] * c;
                part0 += ch == 0 ? p : 0.f;
                part1 += ch == 1 ? p : 0.f;
                part2 += ch == 2 ? p : 0.f;
            }
            part0 = wave_sum(part0);
            part1 = wave_sum(part1);
            part2 = wave_sum(part2);
            if (lane == 0) {
                float bg = white_bg ? (1.0f - acc) : 0.0f;
                rgb_out[ray * 3 + 0] = part0 + bg;
                rgb_out[ray * 3 + 1] = part1 + bg;
                rgb_out[ray * 3 + 2] = part2 + bg;
            }
        }
        if (lane == 0) acc_out[ray] = acc;
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
            a[e] = s < S ? a_row[s] : 0.0f;
            float t = s < S ? (1.0f - a[e] + kEps) : 1.0f;
            float incl = wave_incl_prod(t, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
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
                float c = c_row[k];
                int s = k / 3;
                int ch = k - 3 * s;
                float g = ch == 0 ? g0 : (ch == 1 ? g1 : g2);
                gl[k] = g * (c - wb);
                dc_row[k] = wl[s] * g;
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
                d_alpha[ray * S + s] = G * T[e] - R / t;
            }
        }
    }
}

template <int E>
static int launch_fwd(const float* alpha, const float* rgb, int64_t n_rays, int S, int white_bg, float* weights,
                      float* rgb_out, float* acc_out, hipStream_t st) {
    int64_t blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(composite_fwd_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, rgb, n_rays, S,
                       white_bg, weights, rgb_out, acc_out);
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
