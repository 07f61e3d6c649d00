// Spherical-Gaussian shading over the light-major rows (l, n) -> l*Ns + n, forward and backward.
// Reference: stage2/model/sgbasis.py:16-32 + stage2/model/renderer.py:174-204
//     h = normalize(l + v);  D_k = exp(lambda_k (h.n - 1));  spec_c = max(sum_k w_{c,k} D_k, 0)
//     rgb = clamp((albedo + spec) * I_l * (l.n) * clamp(vis, 0, 1), 0, 1)          (cos is NOT clamped)
// The reference materialises [L*Ns, 27] tiles of the weights, the lobe responses and six more [L*Ns, 3]
// temporaries; here one thread owns a surface point, keeps its albedo / 27 SG weights / normal / view in
// registers and walks the lights, so HBM traffic is the outputs only (24 B per (l, n) forward).  The backward
// recomputes the forward quantities, accumulates the per-point gradients (albedo, weights, normal) in
// registers over the light loop (no atomics) and reduces the per-light gradients (direction, intensity) with
// a wave reduction + one deterministic partial per workgroup.
#include "common.h"

namespace psn {

constexpr int kMaxBasis = 9;

struct ShadeArgs {
    const float* light_dir;   // [L,3]
    const float* view;        // [Ns,3]
    const float* normal;      // [Ns,3]
    const float* albedo;      // [Ns,3]
    const float* weights;     // [Ns, nw] (nw = 3*nb if specular_rgb else nb), already relu'd
    const float* lobe;        // [nb]
    const float* light_int;   // [L] or nullptr (scalar below)
    float light_int_scalar;
    const float* vis;         // [L*Ns] or nullptr
    int L, nb, specular_rgb;
    int64_t Ns;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct PointCtx {
    float v[3], n[3], alb[3];
    float w[3][kMaxBasis];
    float lam[kMaxBasis];
};

__device__ __forceinline__ void load_point(const ShadeArgs& a, int64_t n, bool ok, PointCtx& p) {
    const int nw = a.specular_rgb ? 3 * a.nb : a.nb;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.v[c] = ok ? a.view[n * 3 + c] : 0.f;
        p.n[c] = ok ? a.normal[n * 3 + c] : 0.f;
        p.alb[c] = ok ? a.albedo[n * 3 + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kMaxBasis; ++k) {
        p.lam[k] = k < a.nb ? fmaxf(a.lobe[k], 0.0f) : 0.0f;  // lobe.clamp(min=0), sgbasis.py:25
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float wv = 0.f;
            if (ok && k < a.nb) wv = a.specular_rgb ? a.weights[n * nw + c * a.nb + k] : a.weights[n * nw + k];
            p.w[c][k] = wv;
        }
    }
}

// forward pieces for one (light, point); returns pre-clamp rgb and everything the backward needs
struct Fwd {
    float h[3], inv_norm, hn, D[kMaxBasis], s[3], brdf[3], cosv, vcl, I, pre[3];
};

__device__ __forceinline__ void shade_one(const ShadeArgs& a, const PointCtx& p, const float l[3], float I, float vis,
                                          bool has_vis, Fwd& f) {
    float u0 = l[0] + p.v[0], u1 = l[1] + p.v[1], u2 = l[2] + p.v[2];
    float nrm = sqrtf(u0 * u0 + u1 * u1 + u2 * u2);
    float den = fmaxf(nrm, 1e-12f);  // F.normalize eps
    f.inv_norm = 1.0f / den;
    f.h[0] = u0 / den;
    f.h[1] = u1 / den;
    f.h[2] = u2 / den;
    f.hn = f.h[0] * p.n[0] + f.h[1] * p.n[1] + f.h[2] * p.n[2];
    float t = f.hn - 1.0f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxBasis; ++k) {
        float d = (k < a.nb) ? expf(p.lam[k] * t) : 0.0f;
        f.D[k] = d;
        s0 += p.w[0][k] * d;
        s1 += p.w[1][k] * d;
        s2 += p.w[2][k] * d;
    }
    f.s[0] = s0;
    f.s[1] = a.specular_rgb ? s1 : s0;
    f.s[2] = a.specular_rgb ? s2 : s0;
    f.cosv = l[0] * p.n[0] + l[1] * p.n[1] + l[2] * p.n[2];
    f.vcl = has_vis ? fminf(fmaxf(vis, 0.0f), 1.0f) : 1.0f;
    f.I = I;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f.brdf[c] = p.alb[c] + fmaxf(f.s[c], 0.0f);
        float x = f.brdf[c] * I * f.cosv;
        if (has_vis) x = x * f.vcl;
        f.pre[c] = x;
    }
}

__global__ __launch_bounds__(256) void sg_shade_fwd_kernel(ShadeArgs a, float* __restrict__ rgb, float* __restrict__ spec) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < a.Ns;
    PointCtx p;
    load_point(a, n, ok, p);
    const bool has_vis = a.vis != nullptr;
    const int sc = a.specular_rgb ? 3 : 1;
    for (int l = 0; l < a.L; ++l) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I = a.light_int != nullptr ? a.light_int[l] : a.light_int_scalar;
        const int64_t row = (int64_t)l * a.Ns + n;
        float vis = (has_vis && ok) ? a.vis[row] : 1.0f;
        Fwd f;
        shade_one(a, p, ld, I, vis, has_vis, f);
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[row * 3 + c] = fminf(fmaxf(f.pre[c], 0.0f), 1.0f);
            if (a.specular_rgb) {
#pragma unroll
                for (int c = 0; c < 3; ++c) spec[row * 3 + c] = fmaxf(f.s[c], 0.0f);
            } else {
                spec[row] = fmaxf(f.s[0], 0.0f);
            }
        }
    }
    (void)sc;
}

// d_light_partial [n_blocks, L, 4] = (d_dir xyz, d_intensity) per workgroup
__global__ __launch_bounds__(256) void sg_shade_bwd_kernel(ShadeArgs a, const float* __restrict__ g_rgb,
                                                           const float* __restrict__ g_spec,
                                                           float* __restrict__ d_albedo, float* __restrict__ d_weights,
                                                           float* __restrict__ d_normal, float* __restrict__ d_vis,
                                                           float* __restrict__ d_light_partial) {
    __shared__ float red[4][4];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < a.Ns;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PointCtx p;
    load_point(a, n, ok, p);
    const bool has_vis = a.vis != nullptr;
    float dalb[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
    float dw[3][kMaxBasis];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < kMaxBasis; ++k) dw[c][k] = 0.f;

    for (int l = 0; l < a.L; ++l) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I = a.light_int != nullptr ? a.light_int[l] : a.light_int_scalar;
        const int64_t row = (int64_t)l * a.Ns + n;
        float vis = (has_vis && ok) ? a.vis[row] : 1.0f;
        Fwd f;
        shade_one(a, p, ld, I, vis, has_vis, f);
        float dl[3] = {0.f, 0.f, 0.f}, dI = 0.f, dcos = 0.f, dvcl = 0.f, ds[3];
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float g = g_rgb[row * 3 + c];
                float gp = (f.pre[c] >= 0.0f && f.pre[c] <= 1.0f) ? g : 0.0f;  // clamp(0,1) backward (inclusive)
                float dbrdf = gp * I * f.cosv * f.vcl;
                dI += gp * f.brdf[c] * f.cosv * f.vcl;
                dcos += gp * f.brdf[c] * I * f.vcl;
                dvcl += gp * f.brdf[c] * I * f.cosv;
                dalb[c] += dbrdf;
                float dspec = dbrdf;
                if (g_spec != nullptr) dspec += a.specular_rgb ? g_spec[row * 3 + c] : (c == 0 ? g_spec[row] : 0.0f);
                ds[c] = f.s[c] >= 0.0f ? dspec : 0.0f;  // clamp(min=0) backward
            }
            if (!a.specular_rgb) {  // one shared specular channel
                ds[0] = ds[0] + ds[1] + ds[2];
                ds[1] = 0.f;
                ds[2] = 0.f;
            }
            float dhn = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxBasis; ++k) {
                float dD = ds[0] * p.w[0][k] + ds[1] * p.w[1][k] + ds[2] * p.w[2][k];
                dhn += dD * p.lam[k] * f.D[k];
                dw[0][k] += ds[0] * f.D[k];
                dw[1][k] += ds[1] * f.D[k];
                dw[2][k] += ds[2] * f.D[k];
            }
            // hn = h.n ; cos = l.n
            float dh[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dh[c] = dhn * p.n[c];
                dn[c] += dhn * f.h[c] + dcos * ld[c];
                dl[c] = dcos * p.n[c];
            }
            // h = u / |u|
            float hd = f.h[0] * dh[0] + f.h[1] * dh[1] + f.h[2] * dh[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) dl[c] += (dh[c] - f.h[c] * hd) * f.inv_norm;
            if (d_vis != nullptr) d_vis[row] = (vis >= 0.0f && vis <= 1.0f) ? dvcl : 0.0f;
        }
        // per-light gradients: wave reduce, then one partial per workgroup (deterministic)
        float r0 = wave_sum_f(dl[0]), r1 = wave_sum_f(dl[1]), r2 = wave_sum_f(dl[2]), r3 = wave_sum_f(dI);
        if (lane == 0) {
            red[wave][0] = r0;
            red[wave][1] = r1;
            red[wave][2] = r2;
            red[wave][3] = r3;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            d_light_partial[((int64_t)blockIdx.x * a.L + l) * 4 + threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (ok) {
        const int nw = a.specular_rgb ? 3 * a.nb : a.nb;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d_albedo[n * 3 + c] = dalb[c];
            d_normal[n * 3 + c] = dn[c];
        }
#pragma unroll
        for (int k = 0; k < kMaxBasis; ++k) {
            if (k < a.nb) {
                if (a.specular_rgb) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) d_weights[n * nw + c * a.nb + k] = dw[c][k];
                } else {
                    d_weights[n * nw + k] = dw[0][k];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void sg_light_reduce_kernel(const float* __restrict__ partial, int n_blocks, int L,
                                                              float* __restrict__ d_light_dir,
                                                              float* __restrict__ d_light_int) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over L*4
    if (i >= L * 4) return;
    float s = 0.f;
    for (int b = 0; b < n_blocks; ++b) s += partial[(int64_t)b * L * 4 + i];
    const int l = i >> 2, c = i & 3;
    if (c < 3) d_light_dir[l * 3 + c] = s;
    else if (d_light_int != nullptr) d_light_int[l] = s;
}

static int fill_args(ShadeArgs& a, const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* weights, const float* lobe, const float* light_int, float light_int_scalar,
                     const float* vis, int L, int64_t Ns, int nb, int specular_rgb) {
    PSN_CHECK_ARG(light_dir && view && normal && albedo && weights && lobe, "sg_shade: null pointer");
    PSN_CHECK_ARG(L >= 1 && Ns >= 0 && nb >= 1 && nb <= kMaxBasis, "sg_shade: L=%d nb=%d", L, nb);
    a.light_dir = light_dir; a.view = view; a.normal = normal; a.albedo = albedo; a.weights = weights; a.lobe = lobe;
    a.light_int = light_int; a.light_int_scalar = light_int_scalar; a.vis = vis; a.L = L; a.Ns = Ns; a.nb = nb;
    a.specular_rgb = specular_rgb ? 1 : 0;
    return PSN_OK;
}

}  // namespace psn

extern "C" int psn_sg_shade_fwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                                const float* weights, const float* lobe, const float* light_int,
                                float light_int_scalar, const float* vis, int L, int64_t Ns, int nb, int specular_rgb,
                                float* rgb, float* spec, void* stream) {
    using namespace psn;
    ShadeArgs a;
    int rc = fill_args(a, light_dir, view, normal, albedo, weights, lobe, light_int, light_int_scalar, vis, L, Ns, nb, specular_rgb);
    if (rc) return rc;
    PSN_CHECK_ARG(rgb && spec, "sg_shade_fwd: null output");
    if (Ns == 0) return PSN_OK;
    hipLaunchKernelGGL(sg_shade_fwd_kernel, dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, rgb, spec);
    PSN_CHECK_LAUNCH("sg_shade_fwd");
    return PSN_OK;
}

extern "C" int psn_sg_shade_bwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                                const float* weights, const float* lobe, const float* light_int,
                                float light_int_scalar, const float* vis, int L, int64_t Ns, int nb, int specular_rgb,
                                const float* g_rgb, const float* g_spec, float* d_albedo, float* d_weights,
                                float* d_normal, float* d_vis, float* d_light_dir, float* d_light_int,
                                float* workspace, void* stream) {
    using namespace psn;
    ShadeArgs a;
    int rc = fill_args(a, light_dir, view, normal, albedo, weights, lobe, light_int, light_int_scalar, vis, L, Ns, nb, specular_rgb);
    if (rc) return rc;
    PSN_CHECK_ARG(g_rgb && d_albedo && d_weights && d_normal && d_light_dir && workspace, "sg_shade_bwd: null pointer");
    if (Ns == 0) return PSN_OK;
    const int n_blocks = (int)((Ns + 255) / 256);
    hipLaunchKernelGGL(sg_shade_bwd_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, a, g_rgb, g_spec, d_albedo,
                       d_weights, d_normal, d_vis, workspace);
    PSN_CHECK_LAUNCH("sg_shade_bwd");
    hipLaunchKernelGGL(sg_light_reduce_kernel, dim3((L * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                       n_blocks, L, d_light_dir, d_light_int);
    PSN_CHECK_LAUNCH("sg_shade_bwd light reduce");
    return PSN_OK;
}
