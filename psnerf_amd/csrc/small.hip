// The launch-bound tail of a stage-2 train step as single kernels (each replaces 5-20 elementwise torch launches on
// tensors of 96 ... 30k rows, every one of which costs a dispatch whatever its size):
//   psn_normalize_rows_{fwd,bwd}  F.normalize(x, p=2, dim=-1) of [n, 3] rows and its backward
//                                 (stage2/model/renderer.py:129,137: the predicted normals)
//   psn_light_rows_{fwd,bwd}      the light-table lookups of a step, stage2/trainer.py:376-379: rows l_slt of the direction
//                                 table, normalised, and of the intensity table; backward = dense table gradients
//   psn_camera_rays               stage2/utils/rend_util.py:90-147 (4 x 4 pose): normalised camera rays of selected pixels
//   psn_adam_flat                 torch.optim.Adam's update (stage2/trainer.py:126-133,402-410) over a flat parameter /
//                                 gradient / moment range, torch's arithmetic operation by operation
// All HBM- / latency-bound; nothing here is worth more than a thread per row or element.
#include "common.h"

namespace psn {

// y = x / max(|x|, eps): torch.nn.functional.normalize (norm, clamp_min(eps), div)
__device__ __forceinline__ void normalize3(const float x0, const float x1, const float x2, float eps, float& y0, float& y1, float& y2) {
    const float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    const float d = nrm < eps ? eps : nrm;
    y0 = x0 / d; y1 = x1 / d; y2 = x2 / d;
}
// backward of the above, autograd's chain (div -> clamp_min -> norm): dx = g / d + x / |x| * s, s = -(g . x) / d^2 where the
// clamp is inactive (|x| >= eps), and the norm's subgradient at 0 is 0
__device__ __forceinline__ void normalize3_bwd(const float x0, const float x1, const float x2, const float g0, const float g1,
                                               const float g2, float eps, float& d0, float& d1, float& d2) {
    const float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    const float d = nrm < eps ? eps : nrm;
    d0 = g0 / d; d1 = g1 / d; d2 = g2 / d;
    if (nrm >= eps && nrm > 0.0f) {
        const float s = -((g0 * x0 / d) / d + (g1 * x1 / d) / d + (g2 * x2 / d) / d);
        d0 += x0 / nrm * s; d1 += x1 / nrm * s; d2 += x2 / nrm * s;
    }
}

__global__ __launch_bounds__(256) void normalize_rows_fwd_kernel(const float* __restrict__ x, int64_t n, float eps, float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float y0, y1, y2;
    normalize3(x[3 * i], x[3 * i + 1], x[3 * i + 2], eps, y0, y1, y2);
    y[3 * i] = y0; y[3 * i + 1] = y1; y[3 * i + 2] = y2;
}
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n,
                                                                 float eps, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float d0, d1, d2;
    normalize3_bwd(x[3 * i], x[3 * i + 1], x[3 * i + 2], g[3 * i], g[3 * i + 1], g[3 * i + 2], eps, d0, d1, d2);
    dx[3 * i] = d0; dx[3 * i + 1] = d1; dx[3 * i + 2] = d2;
}

__global__ __launch_bounds__(256) void light_rows_fwd_kernel(const float* __restrict__ dir_tab, const float* __restrict__ int_tab,
                                                             const int64_t* __restrict__ idx, int n_idx, float eps,
                                                             float* __restrict__ dir_out, float* __restrict__ int_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_idx) return;
    const int64_t r = idx[i];
    float y0, y1, y2;
    normalize3(dir_tab[3 * r], dir_tab[3 * r + 1], dir_tab[3 * r + 2], eps, y0, y1, y2);
    dir_out[3 * i] = y0; dir_out[3 * i + 1] = y1; dir_out[3 * i + 2] = y2;
    if (int_out != nullptr) int_out[i] = int_tab[r];
}
// One thread per TABLE row: it scans the step's index list and sums the contributions of its occurrences in list order
// (deterministic for duplicate rows; untouched rows get exact zeros, so no separate fill of the dense gradients).
__global__ __launch_bounds__(256) void light_rows_bwd_kernel(const float* __restrict__ dir_tab, const int64_t* __restrict__ idx, int n_idx,
                                                             int64_t n_rows, float eps, const float* __restrict__ g_dir,
                                                             const float* __restrict__ g_int, float* __restrict__ d_dir,
                                                             float* __restrict__ d_int) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ai = 0.f;
    for (int i = 0; i < n_idx; ++i) {
        if (idx[i] != r) continue;
        if (g_dir != nullptr) {
            float d0, d1, d2;
            normalize3_bwd(dir_tab[3 * r], dir_tab[3 * r + 1], dir_tab[3 * r + 2], g_dir[3 * i], g_dir[3 * i + 1], g_dir[3 * i + 2], eps, d0, d1, d2);
            a0 += d0; a1 += d1; a2 += d2;
        }
        if (g_int != nullptr) ai += g_int[i];
    }
    if (d_dir != nullptr) { d_dir[3 * r] = a0; d_dir[3 * r + 1] = a1; d_dir[3 * r + 2] = a2; }
    if (d_int != nullptr) d_int[r] = ai;
}

// rend_util.py:131-147 + :114-115: x = (u - cx) / fx * 1, y = (v - cy) / fy * 1, d = R [x, y, 1] (sum left to right),
// F.normalize(d); `scale` (+-1) multiplies the result (the shading wants -d: points-to-camera)
__global__ __launch_bounds__(256) void camera_rays_kernel(const float* __restrict__ uv, const float* __restrict__ pose,
                                                          const float* __restrict__ intr, const int64_t* __restrict__ idx, int64_t n,
                                                          float scale, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t p = idx != nullptr ? idx[i] : i;
    const float fx = intr[0], fy = intr[5], cx = intr[2], cy = intr[6];
    const float x = (uv[2 * p] - cx) / fx * 1.0f;
    const float y = (uv[2 * p + 1] - cy) / fy * 1.0f;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = x * pose[4 * c] + y * pose[4 * c + 1] + 1.0f * pose[4 * c + 2];
    float y0, y1, y2;
    normalize3(d[0], d[1], d[2], 1e-12f, y0, y1, y2);
    out[3 * i] = y0 * scale; out[3 * i + 1] = y1 * scale; out[3 * i + 2] = y2 * scale;
}

// torch/optim/adam.py::_multi_tensor_adam (amsgrad off, no weight decay), per element:
//   m = m + w1 (g - m)                     _foreach_lerp_(exp_avgs, grads, 1 - beta1), weight < 0.5
//   v = v * beta2;  v = v + w2 * g * g     _foreach_mul_, _foreach_addcmul_(value = 1 - beta2)
//   den = sqrt(v) / bc2_sqrt + eps         _foreach_sqrt, _foreach_div_, _foreach_add_
//   p = p + neg_step * (m / den)           _foreach_addcdiv_(value = -lr / bias_correction1)
struct AdamSeg { int64_t off, goff, n; float neg_step, bc2_sqrt; };
struct AdamArgs { float* p; const float* g; float* m; float* v; float w1, beta2, w2, eps; AdamSeg seg[PSN_ADAM_MAX_SEGS]; int n_seg; };
__global__ __launch_bounds__(256) void adam_flat_kernel(AdamArgs a) {
    const AdamSeg s = a.seg[blockIdx.y];
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < s.n; i += (int64_t)gridDim.x * 1024) {
        const int64_t e0 = s.off + i, ge0 = s.goff + i;
        const int cnt = s.n - i < 4 ? (int)(s.n - i) : 4;
        if (cnt == 4 && ((e0 | ge0) & 3) == 0) {
            const float4 g4 = *reinterpret_cast<const float4*>(a.g + ge0);
            float4 m4 = *reinterpret_cast<float4*>(a.m + e0), v4 = *reinterpret_cast<float4*>(a.v + e0), p4 = *reinterpret_cast<float4*>(a.p + e0);
            float* gp = (float*)&g4; float* mp = (float*)&m4; float* vp = (float*)&v4; float* pp = (float*)&p4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mp[c] = mp[c] + a.w1 * (gp[c] - mp[c]);
                vp[c] = vp[c] * a.beta2;
                vp[c] = vp[c] + a.w2 * gp[c] * gp[c];
                const float den = sqrtf(vp[c]) / s.bc2_sqrt + a.eps;
                pp[c] = pp[c] + s.neg_step * (mp[c] / den);
            }
            *reinterpret_cast<float4*>(a.m + e0) = m4; *reinterpret_cast<float4*>(a.v + e0) = v4; *reinterpret_cast<float4*>(a.p + e0) = p4;
        } else {
            for (int c = 0; c < cnt; ++c) {
                const float g = a.g[ge0 + c];
                float m = a.m[e0 + c], v = a.v[e0 + c];
                m = m + a.w1 * (g - m);
                v = v * a.beta2;
                v = v + a.w2 * g * g;
                const float den = sqrtf(v) / s.bc2_sqrt + a.eps;
                a.m[e0 + c] = m; a.v[e0 + c] = v;
                a.p[e0 + c] = a.p[e0 + c] + s.neg_step * (m / den);
            }
        }
    }
}

}  // namespace psn

extern "C" int psn_normalize_rows_fwd(const float* x, int64_t n, float eps, float* y, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && y && n >= 0, "normalize_rows_fwd: null pointer");
    if (n == 0) return PSN_OK;
    hipLaunchKernelGGL(normalize_rows_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, eps, y);
    PSN_CHECK_LAUNCH("normalize_rows_fwd");
    return PSN_OK;
}

extern "C" int psn_normalize_rows_bwd(const float* x, const float* g, int64_t n, float eps, float* dx, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && g && dx && n >= 0, "normalize_rows_bwd: null pointer");
    if (n == 0) return PSN_OK;
    hipLaunchKernelGGL(normalize_rows_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, g, n, eps, dx);
    PSN_CHECK_LAUNCH("normalize_rows_bwd");
    return PSN_OK;
}

extern "C" int psn_light_rows_fwd(const float* dir_table, const float* int_table, const int64_t* idx, int n_idx, float eps,
                                  float* dir_out, float* int_out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(dir_table && idx && dir_out && n_idx >= 0 && ((int_table == nullptr) == (int_out == nullptr)), "light_rows_fwd: bad arguments");
    if (n_idx == 0) return PSN_OK;
    hipLaunchKernelGGL(light_rows_fwd_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dir_table, int_table, idx,
                       n_idx, eps, dir_out, int_out);
    PSN_CHECK_LAUNCH("light_rows_fwd");
    return PSN_OK;
}

extern "C" int psn_light_rows_bwd(const float* dir_table, const int64_t* idx, int n_idx, int64_t n_rows, float eps, const float* g_dir,
                                  const float* g_int, float* d_dir_table, float* d_int_table, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(dir_table && idx && n_idx >= 0 && n_rows >= 0, "light_rows_bwd: bad arguments");
    PSN_CHECK_ARG((g_dir == nullptr) == (d_dir_table == nullptr) && (g_int == nullptr) == (d_int_table == nullptr), "light_rows_bwd: gradient pairs");
    if (n_rows == 0) return PSN_OK;
    hipLaunchKernelGGL(light_rows_bwd_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dir_table, idx, n_idx,
                       n_rows, eps, g_dir, g_int, d_dir_table, d_int_table);
    PSN_CHECK_LAUNCH("light_rows_bwd");
    return PSN_OK;
}

extern "C" int psn_camera_rays(const float* uv, const float* pose, const float* intrinsics, const int64_t* idx, int64_t n, float scale,
                               float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(uv && pose && intrinsics && out && n >= 0, "camera_rays: null pointer");
    if (n == 0) return PSN_OK;
    hipLaunchKernelGGL(camera_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, uv, pose, intrinsics, idx, n,
                       scale, out);
    PSN_CHECK_LAUNCH("camera_rays");
    return PSN_OK;
}

extern "C" int psn_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                             float one_minus_beta1, float beta2, float one_minus_beta2, float eps, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && segs && n_segs >= 1 && n_segs <= PSN_ADAM_MAX_SEGS, "adam_flat: bad arguments (n_segs=%d)", n_segs);
    AdamArgs a = {};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.w1 = one_minus_beta1; a.beta2 = beta2; a.w2 = one_minus_beta2; a.eps = eps;
    a.n_seg = n_segs;
    int64_t max_n = 0;
    for (int i = 0; i < n_segs; ++i) {
        PSN_CHECK_ARG(segs[i].offset >= 0 && segs[i].grad_offset >= 0 && segs[i].n >= 0 && segs[i].bias_correction2_sqrt > 0.0f, "adam_flat: segment %d", i);
        a.seg[i].off = segs[i].offset; a.seg[i].goff = segs[i].grad_offset; a.seg[i].n = segs[i].n; a.seg[i].neg_step = segs[i].neg_step_size; a.seg[i].bc2_sqrt = segs[i].bias_correction2_sqrt;
        if (segs[i].n > max_n) max_n = segs[i].n;
    }
    if (max_n == 0) return PSN_OK;
    int64_t blocks = (max_n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks, (unsigned)n_segs), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("adam_flat");
    return PSN_OK;
}
