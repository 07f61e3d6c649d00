// Stage-1 losses and surface normals of the sync-free training forward, a handful of launches instead of ~75:
//   surface normals   stage1/model/rendering.py:200-212: n = g / (|g| + 1e-5) at the surface points and their jittered
//                     neighbours, normal_pred = n where the ray hit, diff_norm = |n - n_neighbour|
//   losses            stage1/model/losses.py:24-70: L1 colour, smoothness (mean of diff_norm over the hit rays), L1 normal
//                     over norm_mask, BCE(acc, mask_gt) over mask_valid -- masked sums over device-resident counts
// All tensors are [N] / [N, 3] with N = rays per step (4096): latency-bound, one thread per ray.
#include "common.h"

namespace psn {

constexpr int kS1Blocks = 64;

struct S1LossArgs {
    const float* rgb; const float* rgb_gt;                                       // [N, 3]
    const float* diff; const unsigned char* hit;                                 // [N] (null: no smoothness term)
    const float* normal; const float* normal_gt; const unsigned char* norm_mask; // [N, 3] x 2, [N] (null: no normal term)
    const float* acc; const float* mask_gt; const unsigned char* mask_valid;     // [N] x 3 (null: no mask term)
    int64_t N;
    float* partial;                                                              // [blocks, 8]
    // backward
    const float* g_loss;        // [1] upstream gradient of the weighted total
    const float* sums;          // [8]: sums[4..6] = hit / norm_mask / mask_valid counts (all-reduced under data parallelism)
    float k_rgb, w_grad, w_norm, w_mask;
    float* d_rgb; float* d_diff; float* d_normal; float* d_acc;
};

__device__ __forceinline__ float sgnf(float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void stage1_loss_fwd_kernel(S1LossArgs a) {
    float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.x * 256) {
        if (a.rgb != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s[0] += fabsf(a.rgb[3 * n + c] - a.rgb_gt[3 * n + c]);
        }
        if (a.hit != nullptr && a.hit[n]) {
            s[4] += 1.0f;
            if (a.diff != nullptr) s[1] += a.diff[n];
        }
        if (a.norm_mask != nullptr && a.norm_mask[n]) {
            s[5] += 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) s[2] += fabsf(a.normal[3 * n + c] - a.normal_gt[3 * n + c]);
        }
        if (a.mask_valid != nullptr && a.mask_valid[n]) {
            s[6] += 1.0f;
            // F.binary_cross_entropy(acc.clamp(0, 1), t): log terms clamped at -100 (ATen binary_cross_entropy_out_cuda)
            const float p = fminf(fmaxf(a.acc[n], 0.0f), 1.0f), t = a.mask_gt[n];
            s[3] += (t - 1.0f) * fmaxf(log1pf(-p), -100.0f) - t * fmaxf(logf(p), -100.0f);
        }
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float v = s[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7)
        a.partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

struct S1Terms { float inv_rays, w_full, w_grad, w_norm, w_mask; int has_grad, has_norm, has_mask; };

// terms[0..3] = colour / smoothness / normal / mask loss, terms[4] = their weighted total (losses.py:38-69, same order)
__device__ __forceinline__ void stage1_terms(const float* sums, const S1Terms& t, float* terms) {
    const float l_rgb = sums[0] * t.inv_rays;
    const float l_grad = t.has_grad ? sums[1] / fmaxf(sums[4], 1.0f) : 0.0f;
    const float l_n = t.has_norm ? sums[2] / fmaxf(sums[5], 1.0f) : 0.0f;
    const float l_m = t.has_mask ? sums[3] / fmaxf(sums[6], 1.0f) : 0.0f;
    float loss = t.w_full * l_rgb + t.w_grad * l_grad;
    if (t.has_norm) loss = loss + t.w_norm * l_n;
    if (t.has_mask) loss = loss + t.w_mask * l_m;
    terms[0] = l_rgb; terms[1] = l_grad; terms[2] = l_n; terms[3] = l_m; terms[4] = loss;
}

// fixed summation order (deterministic); terms == null: only the sums (the caller all-reduces the counts first)
__global__ __launch_bounds__(64) void stage1_loss_reduce_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ sums,
                                                               S1Terms t, float* __restrict__ terms) {
    __shared__ float sh[8];
    if (threadIdx.x < 8) {
        float tot = 0.f;
        if (threadIdx.x < 7)
            for (int b = 0; b < blocks; ++b) tot += partial[(int64_t)b * 8 + threadIdx.x];
        sums[threadIdx.x] = tot;
        sh[threadIdx.x] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0 && terms != nullptr) stage1_terms(sh, t, terms);
}

__global__ void stage1_loss_terms_kernel(const float* __restrict__ sums, S1Terms t, float* __restrict__ terms) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stage1_terms(sums, t, terms);
}

__global__ __launch_bounds__(256) void stage1_loss_bwd_kernel(S1LossArgs a) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const float g = a.g_loss[0];
    if (a.d_rgb != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.d_rgb[3 * n + c] = g * a.k_rgb * sgnf(a.rgb[3 * n + c] - a.rgb_gt[3 * n + c]);
    }
    if (a.d_diff != nullptr) a.d_diff[n] = a.hit[n] ? g * a.w_grad / fmaxf(a.sums[4], 1.0f) : 0.0f;
    if (a.d_normal != nullptr) {
        const bool m = a.norm_mask[n] != 0;
        const float k = g * a.w_norm / fmaxf(a.sums[5], 1.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) a.d_normal[3 * n + c] = m ? k * sgnf(a.normal[3 * n + c] - a.normal_gt[3 * n + c]) : 0.0f;
    }
    if (a.d_acc != nullptr) {
        float d = 0.0f;
        const float x = a.acc[n];
        if (a.mask_valid[n] && x >= 0.0f && x <= 1.0f) {   // clamp's gradient passes on [0, 1]
            // ATen binary_cross_entropy_backward: grad (p - t) / max((1 - p) p, 1e-12)
            const float t = a.mask_gt[n];
            d = g * a.w_mask / fmaxf(a.sums[6], 1.0f) * (x - t) / fmaxf((1.0f - x) * x, 1e-12f);
        }
        a.d_acc[n] = d;
    }
}

// ---- surface normals -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit3(const float* g, float eps, float* n, float& s) {
    s = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const float t = s + eps;
    n[0] = g[0] / t; n[1] = g[1] / t; n[2] = g[2] / t;
}

__global__ __launch_bounds__(256) void surface_normals_fwd_kernel(const float* __restrict__ g, const unsigned char* __restrict__ hit, int64_t N,
                                                                  float eps, float* __restrict__ norm_pred, float* __restrict__ diff) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float a[3], b[3], sa, sb;
    unit3(g + 3 * i, eps, a, sa);
    unit3(g + 3 * (N + i), eps, b, sb);
    const bool h = hit[i] != 0;
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) norm_pred[3 * i + c] = h ? a[c] : 0.0f;
    diff[i] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
}

// dg of n = g / (|g| + eps):  dn / t - g (dn . g) / (t^2 s), the second term 0 at s = 0 (torch's norm backward)
__device__ __forceinline__ void unit3_bwd(const float* g, float eps, const float* dn, float* dg) {
    const float s = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), t = s + eps;
    const float dot = dn[0] * g[0] + dn[1] * g[1] + dn[2] * g[2];
    const float k = s > 0.0f ? dot / (t * t * s) : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dg[c] = dn[c] / t - g[c] * k;
}

__global__ __launch_bounds__(256) void surface_normals_bwd_kernel(const float* __restrict__ g, const unsigned char* __restrict__ hit, int64_t N,
                                                                  float eps, const float* __restrict__ d_norm_pred,
                                                                  const float* __restrict__ d_diff, float* __restrict__ dg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float a[3], b[3], sa, sb, da[3], db[3], o[3];
    unit3(g + 3 * i, eps, a, sa);
    unit3(g + 3 * (N + i), eps, b, sb);
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    const float diff = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float kd = (d_diff != nullptr && diff > 0.0f) ? d_diff[i] / diff : 0.0f;   // d|x| = x / |x|, 0 at x = 0
    const bool h = d_norm_pred != nullptr && hit[i] != 0;
    const float dd[3] = {d0 * kd, d1 * kd, d2 * kd};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        da[c] = (h ? d_norm_pred[3 * i + c] : 0.0f) + dd[c];
        db[c] = -dd[c];
    }
    unit3_bwd(g + 3 * i, eps, da, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) dg[3 * i + c] = o[c];
    unit3_bwd(g + 3 * (N + i), eps, db, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) dg[3 * (N + i) + c] = o[c];
}

}  // namespace psn

extern "C" int psn_stage1_loss_partial_floats(void) { return psn::kS1Blocks * 8; }

static psn::S1Terms s1_terms(int64_t n_rays, const float* weights, bool has_grad, bool has_norm, bool has_mask) {
    psn::S1Terms t;
    t.inv_rays = weights[0] != 0.0f ? 1.0f / (float)n_rays : 0.0f;
    t.w_full = weights[0]; t.w_grad = weights[1]; t.w_norm = weights[2]; t.w_mask = weights[3];
    t.has_grad = has_grad; t.has_norm = has_norm; t.has_mask = has_mask;
    return t;
}

extern "C" int psn_stage1_loss_fwd(const float* rgb, const float* rgb_gt, const float* diff, const unsigned char* hit, const float* normal,
                                   const float* normal_gt, const unsigned char* norm_mask, const float* acc, const float* mask_gt,
                                   const unsigned char* mask_valid, int64_t N, int64_t n_rays, const float* weights, float* partial,
                                   float* sums, float* terms, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(rgb && rgb_gt && weights && partial && sums && N >= 0 && n_rays > 0, "stage1_loss_fwd: null pointer");
    PSN_CHECK_ARG((diff == nullptr || hit) && (normal == nullptr) == (normal_gt == nullptr) && (normal == nullptr) == (norm_mask == nullptr) &&
                  (acc == nullptr) == (mask_gt == nullptr) && (acc == nullptr) == (mask_valid == nullptr), "stage1_loss_fwd: incomplete term");
    S1LossArgs a = {};
    a.rgb = weights[0] != 0.0f ? rgb : nullptr; a.rgb_gt = rgb_gt;
    a.diff = weights[1] != 0.0f ? diff : nullptr; a.hit = hit;
    a.normal = normal; a.normal_gt = normal_gt; a.norm_mask = norm_mask; a.acc = acc; a.mask_gt = mask_gt; a.mask_valid = mask_valid;
    a.N = N; a.partial = partial;
    int bx = (int)((N + 255) / 256);
    bx = bx < 1 ? 1 : (bx > kS1Blocks ? kS1Blocks : bx);
    hipLaunchKernelGGL(stage1_loss_fwd_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("stage1_loss_fwd");
    hipLaunchKernelGGL(stage1_loss_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, bx, sums,
                       s1_terms(n_rays, weights, a.diff != nullptr, normal != nullptr, acc != nullptr), terms);
    PSN_CHECK_LAUNCH("stage1_loss_fwd (reduce)");
    return PSN_OK;
}

extern "C" int psn_stage1_loss_terms(const float* sums, int64_t n_rays, const float* weights, int has_grad, int has_norm, int has_mask,
                                     float* terms, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(sums && weights && terms && n_rays > 0, "stage1_loss_terms: null pointer");
    hipLaunchKernelGGL(stage1_loss_terms_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums,
                       s1_terms(n_rays, weights, has_grad != 0, has_norm != 0, has_mask != 0), terms);
    PSN_CHECK_LAUNCH("stage1_loss_terms");
    return PSN_OK;
}

extern "C" int psn_stage1_loss_bwd(const float* g_loss, const float* sums, const float* rgb, const float* rgb_gt, const unsigned char* hit,
                                   const float* normal, const float* normal_gt, const unsigned char* norm_mask, const float* acc,
                                   const float* mask_gt, const unsigned char* mask_valid, int64_t N, int64_t n_rays, const float* weights,
                                   float* d_rgb, float* d_diff, float* d_normal, float* d_acc, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(g_loss && sums && weights && N >= 0 && n_rays > 0, "stage1_loss_bwd: null pointer");
    PSN_CHECK_ARG((d_rgb == nullptr || (rgb && rgb_gt)) && (d_diff == nullptr || hit) && (d_normal == nullptr || (normal && normal_gt && norm_mask)) &&
                  (d_acc == nullptr || (acc && mask_gt && mask_valid)), "stage1_loss_bwd: a requested gradient lacks its inputs");
    if (N == 0) return PSN_OK;
    S1LossArgs a = {};
    a.rgb = rgb; a.rgb_gt = rgb_gt; a.hit = hit; a.normal = normal; a.normal_gt = normal_gt; a.norm_mask = norm_mask;
    a.acc = acc; a.mask_gt = mask_gt; a.mask_valid = mask_valid; a.N = N; a.g_loss = g_loss; a.sums = sums;
    a.k_rgb = weights[0] / (float)n_rays; a.w_grad = weights[1]; a.w_norm = weights[2]; a.w_mask = weights[3];
    a.d_rgb = d_rgb; a.d_diff = d_diff; a.d_normal = d_normal; a.d_acc = d_acc;
    hipLaunchKernelGGL(stage1_loss_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("stage1_loss_bwd");
    return PSN_OK;
}

extern "C" int psn_surface_normals_fwd(const float* g, const unsigned char* hit, int64_t N, float eps, float* norm_pred, float* diff,
                                       void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(g && hit && norm_pred && diff && N >= 0, "surface_normals_fwd: null pointer");
    if (N == 0) return PSN_OK;
    hipLaunchKernelGGL(surface_normals_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, hit, N, eps,
                       norm_pred, diff);
    PSN_CHECK_LAUNCH("surface_normals_fwd");
    return PSN_OK;
}

extern "C" int psn_surface_normals_bwd(const float* g, const unsigned char* hit, int64_t N, float eps, const float* d_norm_pred,
                                       const float* d_diff, float* dg, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(g && hit && dg && N >= 0, "surface_normals_bwd: null pointer");
    if (N == 0) return PSN_OK;
    hipLaunchKernelGGL(surface_normals_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, hit, N, eps,
                       d_norm_pred, d_diff, dg);
    PSN_CHECK_LAUNCH("surface_normals_bwd");
    return PSN_OK;
}
