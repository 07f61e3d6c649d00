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

// One wave per workgroup: the kernels below give a thread one surface point and loop over the lights, so a launch has
// only Ns threads -- 29,487 at bear.conf sizes = 116 workgroups of 256 threads on 256 CUs.  With 64-thread workgroups the
// same launch spreads over every CU (461 workgroups) and the per-light block reduction is a single wave reduction.
constexpr int kShadeThreads = 64;

constexpr int kMaxBasis = 9;

struct ShadeArgs {
    const float* light_dir;   // [L,3]
    const float* view;        // [Ns,3]
    const float* normal;      // [Ns,3]
    const float* albedo;      // [Ns,3]
    const float* weights;     // [Ns, nw] (nw = 3*nb if specular_rgb else nb), already relu'd
    const float* lobe;        // [nb]
    const float* light_int;   // [L, int_ch] or nullptr (scalar below); int_ch = 3 for RGB envmap lights (eval.py:200)
    int int_ch;
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
    float h[3], inv_norm, hn, D[kMaxBasis], s[3], brdf[3], cosv, vcl, I[3], pre[3];
};

__device__ __forceinline__ void shade_one(const ShadeArgs& a, const PointCtx& p, const float l[3], const float I[3],
                                          float vis, bool has_vis, Fwd& f) {
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
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f.I[c] = I[c];
        f.brdf[c] = p.alb[c] + fmaxf(f.s[c], 0.0f);
        float x = f.brdf[c] * I[c] * f.cosv;
        if (has_vis) x = x * f.vcl;
        f.pre[c] = x;
    }
}

// SG kernels: a workgroup = 64 points x kSgPhases light phases (wave w takes the lights l = w, w + 4, ...): four times the
// waves of the one-thread-per-point form (1844 instead of 461 on 1024 SIMDs at bear.conf sizes), which is what the
// latency of the per-(point, light) arithmetic (9 exponentials, a normalisation) needs.
constexpr int kSgPhases = 4;
__global__ __launch_bounds__(kShadeThreads * kSgPhases) void sg_shade_fwd_kernel(ShadeArgs a, float* __restrict__ rgb, float* __restrict__ spec) {
    const int64_t n = (int64_t)blockIdx.x * kShadeThreads + (threadIdx.x & (kShadeThreads - 1));
    const int phase = threadIdx.x / kShadeThreads;
    const bool ok = n < a.Ns;
    PointCtx p;
    load_point(a, n, ok, p);
    const bool has_vis = a.vis != nullptr;
    const int sc = a.specular_rgb ? 3 : 1;
    for (int l = phase; l < a.L; l += kSgPhases) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            I[c] = a.light_int != nullptr ? a.light_int[l * a.int_ch + (a.int_ch == 3 ? c : 0)] : a.light_int_scalar;
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
__global__ __launch_bounds__(kShadeThreads * kSgPhases) void sg_shade_bwd_kernel(ShadeArgs a, const float* __restrict__ g_rgb,
                                                           const float* __restrict__ g_spec,
                                                           float* __restrict__ d_albedo, float* __restrict__ d_weights,
                                                           float* __restrict__ d_normal, float* __restrict__ d_vis,
                                                           float* __restrict__ d_light_partial) {
    const int64_t n = (int64_t)blockIdx.x * kShadeThreads + (threadIdx.x & (kShadeThreads - 1));
    const int phase = threadIdx.x / kShadeThreads;
    const bool ok = n < a.Ns;
    const int lane = threadIdx.x & 63;
    PointCtx p;
    load_point(a, n, ok, p);
    const bool has_vis = a.vis != nullptr;
    float dalb[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
    float dw[3][kMaxBasis];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < kMaxBasis; ++k) dw[c][k] = 0.f;

    for (int l = phase; l < a.L; l += kSgPhases) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            I[c] = a.light_int != nullptr ? a.light_int[l * a.int_ch + (a.int_ch == 3 ? c : 0)] : a.light_int_scalar;
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
                float dbrdf = gp * I[c] * f.cosv * f.vcl;
                dI += gp * f.brdf[c] * f.cosv * f.vcl;
                dcos += gp * f.brdf[c] * I[c] * f.vcl;
                dvcl += gp * f.brdf[c] * I[c] * f.cosv;
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
        if (lane == 0) *reinterpret_cast<float4*>(d_light_partial + ((int64_t)blockIdx.x * a.L + l) * 4) = make_float4(r0, r1, r2, r3);
    }
    // per-point gradients: the four light phases of a point are summed through LDS in a fixed order (deterministic)
    __shared__ float pred[kSgPhases - 1][6 + 3 * kMaxBasis][kShadeThreads];
    if (phase > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pred[phase - 1][c][lane] = dalb[c];
            pred[phase - 1][3 + c][lane] = dn[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < kMaxBasis; ++k) pred[phase - 1][6 + c * kMaxBasis + k][lane] = dw[c][k];
    }
    __syncthreads();
    if (phase == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dalb[c] = (dalb[c] + pred[0][c][lane]) + (pred[1][c][lane] + pred[2][c][lane]);
            dn[c] = (dn[c] + pred[0][3 + c][lane]) + (pred[1][3 + c][lane] + pred[2][3 + c][lane]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < kMaxBasis; ++k)
                dw[c][k] = (dw[c][k] + pred[0][6 + c * kMaxBasis + k][lane]) + (pred[1][6 + c * kMaxBasis + k][lane] + pred[2][6 + c * kMaxBasis + k][lane]);
    }
    if (ok && phase == 0) {
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
    // 64 outputs (of the L * 4) per workgroup, each summed by four threads over every fourth partial (four independent,
    // pipelined load -> add chains instead of one n_blocks long), combined in a fixed order: deterministic
    __shared__ float red[3][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    if (i < L * 4)
        for (int b = ph; b < n_blocks; b += 4) s += partial[(int64_t)b * L * 4 + i];
    if (ph > 0) red[ph - 1][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph > 0 || i >= L * 4) return;
    s = (s + red[0][threadIdx.x]) + (red[1][threadIdx.x] + red[2][threadIdx.x]);
    const int l = i >> 2, c = i & 3;
    if (c < 3) d_light_dir[l * 3 + c] = s;
    else if (d_light_int != nullptr) d_light_int[l] = s;
}

// ------------------------------------------------------------------------------------------------
// GGX microfacet render model (train.render_model = microfacet): stage2/model/microfacet.py:35-114.
//   lh = normalize(l, 1e-6) etc.;  h = normalize(lh + vh)
//   F = f0 + (1-f0)(1 - lh.h)^5;  a2 = rough^4
//   D = dnn(a2 * [h.n > 0], pi (h.n)^4 (a2 + tan2_m)^2),  tan2_m = dnn(1 - (h.n)^2, (h.n)^2)
//   G = dnn(2 * [dnn(h.v, n.v) > 0], 1 + sqrt(1 + a2 * tan2_v)),  tan2_v = clamp(dnn(1 - c, c), 0), c = clamp((n.v)^2, 0, 1)
//   brdf = dnn(F G D, 4 |l.n| |n.v|) + albedo / pi,   dnn(x, y) = x / (y + 1e-6) with inf/nan -> 0
//   rgb = clamp(brdf * I * (l.n, un-normalised inputs) * clamp(vis, 0, 1), 0, 1)
// Same thread-per-point / loop-over-lights structure as the SG kernels; the backward carries the analytic
// partials of the scalar chain (h.n, n.v, l.h, l.n, a2) and the three normalisations.
struct MfArgs {
    const float* light_dir;  // [L,3]
    const float* view;       // [Ns,3]
    const float* normal;     // [Ns,3]
    const float* albedo;     // [Ns,3]
    const float* rough;      // [Ns]
    const float* light_int;  // [L] or nullptr
    float light_int_scalar, f0;
    const float* vis;        // [L*Ns] or nullptr
    int L;
    int64_t Ns;
};

__device__ __forceinline__ float dnn(float x, float y) {
    float a = x / (y + 1e-6f);
    return (isinf(a) || isnan(a)) ? 0.0f : a;
}
__device__ __forceinline__ bool dnn_ok(float x, float y) {
    float a = x / (y + 1e-6f);
    return !(isinf(a) || isnan(a));
}
__device__ __forceinline__ void normalize_eps(const float x[3], float eps, float out[3], float& inv) {
    float n = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float d = fmaxf(n, eps);
    inv = 1.0f / d;
    out[0] = x[0] / d;
    out[1] = x[1] / d;
    out[2] = x[2] / d;
}
// adjoint of y = x / max(|x|, eps) (constant denominator below eps, like F.normalize)
__device__ __forceinline__ void normalize_bwd(const float y[3], float inv, bool clamped, const float dy[3], float dx[3]) {
    if (clamped) {
        dx[0] = dy[0] * inv; dx[1] = dy[1] * inv; dx[2] = dy[2] * inv;
    } else {
        float yd = y[0] * dy[0] + y[1] * dy[1] + y[2] * dy[2];
        dx[0] = (dy[0] - y[0] * yd) * inv;
        dx[1] = (dy[1] - y[1] * yd) * inv;
        dx[2] = (dy[2] - y[2] * yd) * inv;
    }
}

struct MfFwd {
    float lh[3], inv_l, hh[3], inv_h, cm, cv, lhh, ln, a2, F, D, G, fgd, den2, mf, cosv, vcl;
    float t_m, den_d, t_v_raw, t_v, sq, den_g, chi_d, chi_g;
    bool cl_l, cl_h, ok_tm, ok_d, ok_tv, ok_g, ok_mf;
};

__device__ __forceinline__ void mf_one(const float l[3], const float vh[3], const float nh[3], const float nraw[3],
                                       float rough, float f0, float vis, bool has_vis, MfFwd& f) {
    const float PI = 3.14159265358979323846f;
    normalize_eps(l, 1e-6f, f.lh, f.inv_l);
    f.cl_l = sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]) < 1e-6f;
    float u[3] = {f.lh[0] + vh[0], f.lh[1] + vh[1], f.lh[2] + vh[2]};
    normalize_eps(u, 1e-6f, f.hh, f.inv_h);
    f.cl_h = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) < 1e-6f;
    f.lhh = f.lh[0] * f.hh[0] + f.lh[1] * f.hh[1] + f.lh[2] * f.hh[2];
    float om = 1.0f - f.lhh;
    f.F = f0 + (1.0f - f0) * (om * om * om * om * om);
    float alpha = rough * rough;
    f.a2 = alpha * alpha;
    // D
    f.cm = f.hh[0] * nh[0] + f.hh[1] * nh[1] + f.hh[2] * nh[2];
    f.chi_d = f.cm > 0.0f ? 1.0f : 0.0f;
    float cm2 = f.cm * f.cm;
    f.ok_tm = dnn_ok(1.0f - cm2, cm2);
    f.t_m = dnn(1.0f - cm2, cm2);
    float q = f.a2 + f.t_m;
    f.den_d = PI * (cm2 * cm2) * (q * q);
    f.ok_d = dnn_ok(f.a2 * f.chi_d, f.den_d);
    f.D = dnn(f.a2 * f.chi_d, f.den_d);
    // G
    f.cv = nh[0] * vh[0] + nh[1] * vh[1] + nh[2] * vh[2];
    float hv = f.hh[0] * vh[0] + f.hh[1] * vh[1] + f.hh[2] * vh[2];
    f.chi_g = dnn(hv, f.cv) > 0.0f ? 1.0f : 0.0f;
    float cv2 = fminf(fmaxf(f.cv * f.cv, 0.0f), 1.0f);
    f.ok_tv = dnn_ok(1.0f - cv2, cv2);
    f.t_v_raw = dnn(1.0f - cv2, cv2);
    f.t_v = fmaxf(f.t_v_raw, 0.0f);
    f.sq = sqrtf(1.0f + f.a2 * f.t_v);
    f.den_g = 1.0f + f.sq;
    f.ok_g = dnn_ok(f.chi_g * 2.0f, f.den_g);
    f.G = dnn(f.chi_g * 2.0f, f.den_g);
    f.ln = f.lh[0] * nh[0] + f.lh[1] * nh[1] + f.lh[2] * nh[2];
    f.den2 = 4.0f * fabsf(f.ln) * fabsf(f.cv);
    f.fgd = f.F * f.G * f.D;
    f.ok_mf = dnn_ok(f.fgd, f.den2);
    f.mf = dnn(f.fgd, f.den2);
    f.cosv = l[0] * nraw[0] + l[1] * nraw[1] + l[2] * nraw[2];
    f.vcl = has_vis ? fminf(fmaxf(vis, 0.0f), 1.0f) : 1.0f;
}

__global__ __launch_bounds__(kShadeThreads) void mf_shade_fwd_kernel(MfArgs a, float* __restrict__ rgb) {
    const int64_t n = (int64_t)blockIdx.x * kShadeThreads + threadIdx.x;
    const bool ok = n < a.Ns;
    float v[3], nr[3], alb[3], vh[3], nh[3], inv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[c] = ok ? a.view[n * 3 + c] : 1.f;
        nr[c] = ok ? a.normal[n * 3 + c] : 1.f;
        alb[c] = ok ? a.albedo[n * 3 + c] : 0.f;
    }
    const float rough = ok ? a.rough[n] : 0.5f;
    normalize_eps(v, 1e-6f, vh, inv);
    normalize_eps(nr, 1e-6f, nh, inv);
    const bool has_vis = a.vis != nullptr;
    for (int l = 0; l < a.L; ++l) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I = a.light_int != nullptr ? a.light_int[l] : a.light_int_scalar;
        const int64_t row = (int64_t)l * a.Ns + n;
        float vis = (has_vis && ok) ? a.vis[row] : 1.0f;
        MfFwd f;
        mf_one(ld, vh, nh, nr, rough, a.f0, vis, has_vis, f);
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x = (f.mf + alb[c] / 3.14159265358979323846f) * I * f.cosv;
                if (has_vis) x = x * f.vcl;
                rgb[row * 3 + c] = fminf(fmaxf(x, 0.0f), 1.0f);
            }
        }
    }
}

__global__ __launch_bounds__(kShadeThreads) void mf_shade_bwd_kernel(MfArgs a, const float* __restrict__ g_rgb,
                                                           float* __restrict__ d_albedo, float* __restrict__ d_rough,
                                                           float* __restrict__ d_normal, float* __restrict__ d_vis,
                                                           float* __restrict__ d_light_partial) {
    const float PI = 3.14159265358979323846f;
    const int64_t n = (int64_t)blockIdx.x * kShadeThreads + threadIdx.x;
    const bool ok = n < a.Ns;
    const int lane = threadIdx.x & 63;
    float v[3], nr[3], alb[3], vh[3], nh[3], inv_v, inv_n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[c] = ok ? a.view[n * 3 + c] : 1.f;
        nr[c] = ok ? a.normal[n * 3 + c] : 1.f;
        alb[c] = ok ? a.albedo[n * 3 + c] : 0.f;
    }
    const float rough = ok ? a.rough[n] : 0.5f;
    normalize_eps(v, 1e-6f, vh, inv_v);
    normalize_eps(nr, 1e-6f, nh, inv_n);
    const bool cl_n = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]) < 1e-6f;
    const bool has_vis = a.vis != nullptr;
    float dalb[3] = {0.f, 0.f, 0.f}, dnraw[3] = {0.f, 0.f, 0.f}, dnh[3] = {0.f, 0.f, 0.f}, da2 = 0.f;
    for (int l = 0; l < a.L; ++l) {
        float ld[3] = {a.light_dir[l * 3 + 0], a.light_dir[l * 3 + 1], a.light_dir[l * 3 + 2]};
        float I = a.light_int != nullptr ? a.light_int[l] : a.light_int_scalar;
        const int64_t row = (int64_t)l * a.Ns + n;
        float vis = (has_vis && ok) ? a.vis[row] : 1.0f;
        MfFwd f;
        mf_one(ld, vh, nh, nr, rough, a.f0, vis, has_vis, f);
        float dl[3] = {0.f, 0.f, 0.f}, dI = 0.f;
        if (ok) {
            float dmf = 0.f, dcos = 0.f, dvcl = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float brdf = f.mf + alb[c] / PI;
                float pre = brdf * I * f.cosv;
                if (has_vis) pre = pre * f.vcl;
                float g = g_rgb[row * 3 + c];
                float gp = (pre >= 0.0f && pre <= 1.0f) ? g : 0.0f;
                float dbrdf = gp * I * f.cosv * f.vcl;
                dI += gp * brdf * f.cosv * f.vcl;
                dcos += gp * brdf * I * f.vcl;
                dvcl += gp * brdf * I * f.cosv;
                dalb[c] += dbrdf / PI;
                dmf += dbrdf;
            }
            // mf = fgd / (den2 + eps)
            float dfgd = 0.f, dden2 = 0.f;
            if (f.ok_mf) {
                float r = 1.0f / (f.den2 + 1e-6f);
                dfgd = dmf * r;
                dden2 = -dmf * f.fgd * r * r;
            }
            float dF = dfgd * f.G * f.D, dG = dfgd * f.F * f.D, dD = dfgd * f.F * f.G;
            float sgn_ln = f.ln > 0.f ? 1.f : (f.ln < 0.f ? -1.f : 0.f);
            float sgn_cv = f.cv > 0.f ? 1.f : (f.cv < 0.f ? -1.f : 0.f);
            float dln = dden2 * 4.0f * sgn_ln * fabsf(f.cv);
            float dcv = dden2 * 4.0f * fabsf(f.ln) * sgn_cv;
            float dlhh = dF * (-5.0f) * (1.0f - a.f0) * ((1.0f - f.lhh) * (1.0f - f.lhh) * (1.0f - f.lhh) * (1.0f - f.lhh));
            // D = a2 chi / (den_d + eps)
            float dcm = 0.f, da2_l = 0.f;
            if (f.ok_d) {
                float r = 1.0f / (f.den_d + 1e-6f);
                float dden_d = -dD * f.a2 * f.chi_d * r * r;
                da2_l += dD * f.chi_d * r;
                float cm2 = f.cm * f.cm;
                float q = f.a2 + f.t_m;
                // den_d = pi cm2^2 q^2
                float dq = dden_d * PI * cm2 * cm2 * 2.0f * q;
                float dcm2 = dden_d * PI * 2.0f * cm2 * q * q;
                da2_l += dq;
                if (f.ok_tm) {  // t_m = (1 - cm2) / (cm2 + eps)
                    float rr = 1.0f / (cm2 + 1e-6f);
                    dcm2 += dq * (-(rr) - (1.0f - cm2) * rr * rr);
                }
                dcm += dcm2 * 2.0f * f.cm;
            }
            // G = 2 chi_g / (den_g + eps),  den_g = 1 + sqrt(1 + a2 t_v)
            if (f.ok_g) {
                float r = 1.0f / (f.den_g + 1e-6f);
                float dden_g = -dG * 2.0f * f.chi_g * r * r;
                float dsq = dden_g;
                float dinner = f.sq > 0.f ? dsq * 0.5f / f.sq : 0.f;
                da2_l += dinner * f.t_v;
                float dtv = dinner * f.a2;
                if (f.t_v_raw >= 0.0f && f.ok_tv) {  // clamp(min=0) passes the gradient at >= 0
                    float c2 = f.cv * f.cv;
                    if (c2 >= 0.0f && c2 <= 1.0f) {
                        float cv2 = fminf(fmaxf(c2, 0.0f), 1.0f);
                        float rr = 1.0f / (cv2 + 1e-6f);
                        float dcv2 = dtv * (-(rr) - (1.0f - cv2) * rr * rr);
                        dcv += dcv2 * 2.0f * f.cv;
                    }
                }
            }
            da2 += da2_l;
            // scalar products -> vectors
            float dhh[3], dlh[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dnh[c] += dcm * f.hh[c] + dcv * vh[c] + dln * f.lh[c];
                dhh[c] = dcm * nh[c] + dlhh * f.lh[c];
                dlh[c] = dlhh * f.hh[c] + dln * nh[c];
                dnraw[c] += dcos * ld[c];
                dl[c] = dcos * nr[c];
            }
            float du[3];
            normalize_bwd(f.hh, f.inv_h, f.cl_h, dhh, du);
#pragma unroll
            for (int c = 0; c < 3; ++c) dlh[c] += du[c];
            float dlraw[3];
            normalize_bwd(f.lh, f.inv_l, f.cl_l, dlh, dlraw);
#pragma unroll
            for (int c = 0; c < 3; ++c) dl[c] += dlraw[c];
            if (d_vis != nullptr) d_vis[row] = (vis >= 0.0f && vis <= 1.0f) ? dvcl : 0.0f;
        }
        float r0 = wave_sum_f(dl[0]), r1 = wave_sum_f(dl[1]), r2 = wave_sum_f(dl[2]), r3 = wave_sum_f(dI);
        if (lane == 0) *reinterpret_cast<float4*>(d_light_partial + ((int64_t)blockIdx.x * a.L + l) * 4) = make_float4(r0, r1, r2, r3);
    }
    if (ok) {
        float dn_from_nh[3];
        normalize_bwd(nh, inv_n, cl_n, dnh, dn_from_nh);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d_albedo[n * 3 + c] = dalb[c];
            d_normal[n * 3 + c] = dnraw[c] + dn_from_nh[c];
        }
        d_rough[n] = da2 * 4.0f * rough * rough * rough;  // a2 = rough^4
    }
}

static int fill_args(ShadeArgs& a, const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* weights, const float* lobe, const float* light_int, int int_ch, float light_int_scalar,
                     const float* vis, int L, int64_t Ns, int nb, int specular_rgb) {
    PSN_CHECK_ARG(light_dir && view && normal && albedo && weights && lobe, "sg_shade: null pointer");
    PSN_CHECK_ARG(L >= 1 && Ns >= 0 && nb >= 1 && nb <= kMaxBasis, "sg_shade: L=%d nb=%d", L, nb);
    a.light_dir = light_dir; a.view = view; a.normal = normal; a.albedo = albedo; a.weights = weights; a.lobe = lobe;
    PSN_CHECK_ARG(int_ch == 1 || int_ch == 3, "sg_shade: light intensity must have 1 or 3 channels");
    a.light_int = light_int; a.int_ch = int_ch; a.light_int_scalar = light_int_scalar; a.vis = vis; a.L = L; a.Ns = Ns; a.nb = nb;
    a.specular_rgb = specular_rgb ? 1 : 0;
    return PSN_OK;
}

}  // namespace psn

extern "C" int psn_sg_shade_fwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                                const float* weights, const float* lobe, const float* light_int, int int_ch,
                                float light_int_scalar, const float* vis, int L, int64_t Ns, int nb, int specular_rgb,
                                float* rgb, float* spec, void* stream) {
    using namespace psn;
    ShadeArgs a;
    int rc = fill_args(a, light_dir, view, normal, albedo, weights, lobe, light_int, int_ch, light_int_scalar, vis, L, Ns, nb, specular_rgb);
    if (rc) return rc;
    PSN_CHECK_ARG(rgb && spec, "sg_shade_fwd: null output");
    if (Ns == 0) return PSN_OK;
    hipLaunchKernelGGL(sg_shade_fwd_kernel, dim3((unsigned)((Ns + kShadeThreads - 1) / kShadeThreads)), dim3(kShadeThreads * kSgPhases), 0, (hipStream_t)stream, a, rgb, spec);
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
    int rc = fill_args(a, light_dir, view, normal, albedo, weights, lobe, light_int, 1, light_int_scalar, vis, L, Ns, nb, specular_rgb);
    if (rc) return rc;
    PSN_CHECK_ARG(g_rgb && d_albedo && d_weights && d_normal && d_light_dir && workspace, "sg_shade_bwd: null pointer");
    if (Ns == 0) return PSN_OK;
    const int n_blocks = (int)((Ns + kShadeThreads - 1) / kShadeThreads);
    hipLaunchKernelGGL(sg_shade_bwd_kernel, dim3(n_blocks), dim3(kShadeThreads * kSgPhases), 0, (hipStream_t)stream, a, g_rgb, g_spec, d_albedo,
                       d_weights, d_normal, d_vis, workspace);
    PSN_CHECK_LAUNCH("sg_shade_bwd");
    hipLaunchKernelGGL(sg_light_reduce_kernel, dim3((L * 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, workspace,
                       n_blocks, L, d_light_dir, d_light_int);
    PSN_CHECK_LAUNCH("sg_shade_bwd light reduce");
    return PSN_OK;
}

extern "C" int psn_mf_shade_fwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                                const float* rough, const float* light_int, float light_int_scalar, float f0,
                                const float* vis, int L, int64_t Ns, float* rgb, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(light_dir && view && normal && albedo && rough && rgb, "mf_shade_fwd: null pointer");
    PSN_CHECK_ARG(L >= 1 && Ns >= 0, "mf_shade_fwd: L=%d", L);
    if (Ns == 0) return PSN_OK;
    MfArgs a{light_dir, view, normal, albedo, rough, light_int, light_int_scalar, f0, vis, L, Ns};
    hipLaunchKernelGGL(mf_shade_fwd_kernel, dim3((unsigned)((Ns + kShadeThreads - 1) / kShadeThreads)), dim3(kShadeThreads), 0, (hipStream_t)stream, a, rgb);
    PSN_CHECK_LAUNCH("mf_shade_fwd");
    return PSN_OK;
}

extern "C" int psn_mf_shade_bwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                                const float* rough, const float* light_int, float light_int_scalar, float f0,
                                const float* vis, int L, int64_t Ns, const float* g_rgb, float* d_albedo,
                                float* d_rough, float* d_normal, float* d_vis, float* d_light_dir, float* d_light_int,
                                float* workspace, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(light_dir && view && normal && albedo && rough && g_rgb && d_albedo && d_rough && d_normal &&
                  d_light_dir && workspace, "mf_shade_bwd: null pointer");
    PSN_CHECK_ARG(L >= 1 && Ns >= 0, "mf_shade_bwd: L=%d", L);
    if (Ns == 0) return PSN_OK;
    MfArgs a{light_dir, view, normal, albedo, rough, light_int, light_int_scalar, f0, vis, L, Ns};
    const int n_blocks = (int)((Ns + kShadeThreads - 1) / kShadeThreads);
    hipLaunchKernelGGL(mf_shade_bwd_kernel, dim3(n_blocks), dim3(kShadeThreads), 0, (hipStream_t)stream, a, g_rgb, d_albedo, d_rough,
                       d_normal, d_vis, workspace);
    PSN_CHECK_LAUNCH("mf_shade_bwd");
    hipLaunchKernelGGL(sg_light_reduce_kernel, dim3((L * 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, workspace,
                       n_blocks, L, d_light_dir, d_light_int);
    PSN_CHECK_LAUNCH("mf_shade_bwd light reduce");
    return PSN_OK;
}
