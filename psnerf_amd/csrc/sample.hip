// Sample points along rays: the depth profiles of stage1/model/rendering.py and the points they generate, in one
// launch per ray group instead of ~40 elementwise torch launches (all of them latency-bound at a few thousand rays).
//
//   miss rays / ray-march sweep  (rendering.py:150-162, 431-436):  d = near (1 - u) + far_n u,  u = linspace(0, 1, S)
//   hit rays                     (rendering.py:110-149, 163-176):  interval [dnp, dfp] = [max(d_n - delta, near),
//                                  min(d_n + delta, far_n)] with `steps` samples, preceded (it > 5000) by `steps_out`
//                                  samples of [near, dnp]; the reference sorts the concatenation: the identity unless
//                                  an interval collapses, see the kernel
//   jitter                       (rendering.py:133-141):  d_i <- lo_i + (hi_i - lo_i) noise_i, mid-point bounds
//   points                       p = origin_n + direction_n d
// HBM-bound: 12 S bytes written per ray (+ 4 S of noise read).  Arithmetic mirrors the reference's op order (products
// and sums rounded separately, -ffp-contract=off), so the result is bit-identical to the torch formulation for the
// same u / 1-u tables and noise.
#include "common.h"

namespace psn {

struct SampleArgs {
    const float* origin;   // [N,3]
    const float* dir;      // [N,3]
    const float* dist;     // [N] surface depth (hit rays) or nullptr
    const float* far;      // [N] sphere exit depth
    const int64_t* idx;    // [n] rays of this group (rows of the [N,S,3] output) or nullptr = all rays in order
    const float* u0; const float* omu0;  // [c0] linspace(0,1,c0) and 1 - it
    const float* u1; const float* omu1;  // [c1] or nullptr
    const float* noise;    // [n, c0 + c1] or nullptr
    const unsigned char* flags;          // [N] per-ray group (1 = hit profile, 0 = miss profile) or nullptr = `hit` for all
    const float* um; const float* omum;  // [c0 + c1] linspace(0,1) / 1 - it: the miss profile when flags != nullptr
    float* out;            // [N, c0 + c1, 3]
    int64_t n;
    int c0, c1, hit;
    float near, delta;
};

// depth number s of the group's profile for one ray
__device__ __forceinline__ float profile_depth(const SampleArgs& a, int s, float lo0, float hi0, float lo1, float hi1) {
    if (s < a.c0) return lo0 * a.omu0[s] + hi0 * a.u0[s];
    return lo1 * a.omu1[s - a.c0] + hi1 * a.u1[s - a.c0];
}

__global__ __launch_bounds__(256) void sample_points_kernel(SampleArgs a) {
    const int S = a.c0 + a.c1;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n * S) return;
    const int64_t r = e / S;
    const int s = (int)(e % S);
    const int64_t ray = a.idx != nullptr ? a.idx[r] : r;
    float lo0, hi0, lo1 = 0.f, hi1 = 0.f;
    // per-ray group flags: both ray groups of a step in ONE launch without index lists (which would have to come from a
    // nonzero() = a host synchronisation); a miss ray then reads the single-segment table um / omum
    const bool is_hit = a.flags != nullptr ? a.flags[ray] != 0 : a.hit != 0;
    const bool miss_tab = a.flags != nullptr && !is_hit;
    if (is_hit) {
        const float d = a.dist[ray];
        float dnp = d - a.delta, dfp = d + a.delta;
        dnp = dnp < a.near ? a.near : dnp;
        const float fr = a.far[ray];
        dfp = dfp > fr ? fr : dfp;
        if (a.c1 > 0) { lo0 = a.near; hi0 = dnp; lo1 = dnp; hi1 = dfp; }   // [near, dnp] then [dnp, dfp]
        else { lo0 = dnp; hi0 = dfp; }
    } else {
        lo0 = a.near; hi0 = a.far[ray];
    }
    // The reference SORTS the concatenated outer + inner depths (rendering.py:129).  That is the identity whenever the
    // sequence is non-decreasing -- always, except when an interval collapses (surface within delta of the near plane:
    // near (1 - u) + near u wobbles by an ulp) -- so the sequence is scanned once and only a non-monotonic ray pays for
    // a rank-based selection of its sorted depths.
    bool mono = true;
    if (a.c1 > 0 && !miss_tab) {
        float prev = profile_depth(a, 0, lo0, hi0, lo1, hi1);
        for (int i = 1; i < S; ++i) {
            const float cur = profile_depth(a, i, lo0, hi0, lo1, hi1);
            mono = mono && (cur >= prev);
            prev = cur;
        }
    }
    auto depth_at = [&](int pos) -> float {
        if (miss_tab) return lo0 * a.omum[pos] + hi0 * a.um[pos];
        if (mono) return profile_depth(a, pos, lo0, hi0, lo1, hi1);
        for (int j = 0; j < S; ++j) {  // the element whose (stable) rank is pos
            const float vj = profile_depth(a, j, lo0, hi0, lo1, hi1);
            int rank = 0;
            for (int i = 0; i < S; ++i) {
                const float vi = profile_depth(a, i, lo0, hi0, lo1, hi1);
                rank += (vi < vj || (vi == vj && i < j)) ? 1 : 0;
            }
            if (rank == pos) return vj;
        }
        return profile_depth(a, pos, lo0, hi0, lo1, hi1);
    };
    float d = depth_at(s);
    if (a.noise != nullptr) {
        const float dm = s > 0 ? depth_at(s - 1) : d;
        const float dp = s + 1 < S ? depth_at(s + 1) : d;
        const float lo = s > 0 ? 0.5f * (d + dm) : d;
        const float hi = s + 1 < S ? 0.5f * (dp + d) : d;
        d = lo + (hi - lo) * a.noise[r * S + s];
    }
    const float* o = a.origin + ray * 3;
    const float* v = a.dir + ray * 3;
    float* p = a.out + (ray * S + s) * 3;
    p[0] = o[0] + v[0] * d;
    p[1] = o[1] + v[1] * d;
    p[2] = o[2] + v[2] * d;
}

}  // namespace psn

static int sample_points_impl(const float* origin, const float* dir, const float* dist, const float* far, const int64_t* idx,
                              int64_t n, int hit, float near, float delta, const float* u0, const float* omu0, int c0,
                              const float* u1, const float* omu1, int c1, const float* noise, const unsigned char* flags,
                              const float* um, const float* omum, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(origin && dir && far && out && u0 && omu0 && c0 >= 1, "sample_points: null pointer or empty profile");
    PSN_CHECK_ARG(!hit || dist, "sample_points: hit rays need their surface depth");
    PSN_CHECK_ARG(c1 == 0 || (hit && u1 && omu1 && c1 > 0), "sample_points: the second segment belongs to hit rays");
    if (n <= 0) return PSN_OK;
    SampleArgs a;
    a.origin = origin; a.dir = dir; a.dist = dist; a.far = far; a.idx = idx; a.u0 = u0; a.omu0 = omu0; a.u1 = u1; a.omu1 = omu1;
    a.noise = noise; a.out = out; a.n = n; a.c0 = c0; a.c1 = c1; a.hit = hit; a.near = near; a.delta = delta;
    a.flags = flags; a.um = um; a.omum = omum;
    PSN_CHECK_ARG(flags == nullptr || (um && omum && dist && idx == nullptr), "sample_points: per-ray flags need the miss table, dist and no index list");
    const int64_t total = n * (int64_t)(c0 + c1);
    const int64_t blocks = (total + 255) / 256;
    PSN_CHECK_ARG(blocks < (1ll << 31), "sample_points: too many samples");
    hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("sample_points");
    return PSN_OK;
}

extern "C" int psn_sample_points(const float* origin, const float* dir, const float* dist, const float* far, const int64_t* idx,
                                 int64_t n, int hit, float near, float delta, const float* u0, const float* omu0, int c0,
                                 const float* u1, const float* omu1, int c1, const float* noise, float* out, void* stream) {
    return sample_points_impl(origin, dir, dist, far, idx, n, hit, near, delta, u0, omu0, c0, u1, omu1, c1, noise, nullptr, nullptr,
                              nullptr, out, stream);
}

extern "C" int psn_sample_points_flagged(const float* origin, const float* dir, const float* dist, const float* far,
                                         const unsigned char* flags, int64_t n, float near, float delta, const float* u0,
                                         const float* omu0, int c0, const float* u1, const float* omu1, int c1, const float* um,
                                         const float* omum, const float* noise, float* out, void* stream) {
    PSN_CHECK_ARG(flags != nullptr, "sample_points_flagged: null flags");
    return sample_points_impl(origin, dir, dist, far, nullptr, n, 1, near, delta, u0, omu0, c0, u1, omu1, c1, noise, flags, um, omum,
                              out, stream);
}

// ---- secant refinement step (stage1/model/rendering.py:525-555) ------------------------------------------------------
// One regula-falsi iteration for every hit ray in one launch instead of ~14 elementwise torch launches:
//   f_mid = occ - tau;  the bracket end on f_mid's side moves to d_pred;  d_pred = -f_low (d_high - d_low) / (f_high - f_low) + d_low;
//   p_mid = origin + d_pred * direction                (the query point of the NEXT occupancy evaluation)
// occ == nullptr: initial step (only d_pred and p_mid from the given bracket).  Arithmetic in the reference's op order.
namespace psn {
__global__ __launch_bounds__(256) void secant_step_kernel(const float* __restrict__ occ, float tau, float* __restrict__ d_pred,
                                                          float* __restrict__ d_low, float* __restrict__ d_high,
                                                          float* __restrict__ f_low, float* __restrict__ f_high,
                                                          const float* __restrict__ origin, const float* __restrict__ dir,
                                                          float* __restrict__ p_mid, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float dl = d_low[i], dh = d_high[i], fl = f_low[i], fh = f_high[i];
    if (occ != nullptr) {
        const float fm = occ[i] - tau;
        const float dp = d_pred[i];
        if (fm < 0.0f) { dl = dp; fl = fm; } else { dh = dp; fh = fm; }
        d_low[i] = dl; d_high[i] = dh; f_low[i] = fl; f_high[i] = fh;
    }
    const float dp = (-fl) * (dh - dl) / (fh - fl) + dl;
    d_pred[i] = dp;
    if (p_mid != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p_mid[3 * i + c] = origin[3 * i + c] + dp * dir[3 * i + c];
    }
}
}  // namespace psn

extern "C" int psn_secant_step(const float* occ, float tau, float* d_pred, float* d_low, float* d_high, float* f_low,
                               float* f_high, const float* origin, const float* dir, float* p_mid, int64_t n, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(d_pred && d_low && d_high && f_low && f_high, "secant_step: null pointer");
    PSN_CHECK_ARG(p_mid == nullptr || (origin && dir), "secant_step: p_mid needs origin and dir");
    if (n <= 0) return PSN_OK;
    hipLaunchKernelGGL(secant_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, occ, tau, d_pred,
                       d_low, d_high, f_low, f_high, origin, dir, p_mid, n);
    PSN_CHECK_LAUNCH("secant_step");
    return PSN_OK;
}

// ---- first free -> occupied crossing of the ray-march sweep (stage1/model/rendering.py:457-504) ----------------------
// One wave per ray over its M sweep values val[m] = occupancy - tau:
//   first_free = val[0] < 0;  i* = the first m with val[m] val[m+1] < 0 (the reference takes argmin of
//   sign(val[m] val[m+1]) (M - m), whose unique minimum is that m whenever one exists);  the crossing counts if it goes
//   from free to occupied (val[i*] < 0) and the ray starts in free space;  bracket = sweep depths / values i*, i* + 1.
// Outputs: bracket [4, N] (d_low, d_high, f_low, f_high; a benign (0, 1, -1, 1) for rays without a crossing),
// flags [N] int32: bit 0 = crossing found (mask), bit 1 = first_free.
namespace psn {
__global__ __launch_bounds__(256) void first_crossing_kernel(const float* __restrict__ occ, const float* __restrict__ far,
                                                             const float* __restrict__ u, const float* __restrict__ omu,
                                                             float near, float tau, int64_t n, int M,
                                                             float* __restrict__ bracket, int* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float* v = occ + ray * M;
    int best = M;  // first index with a sign change
    for (int m = lane; m + 1 < M; m += 64) {
        const float p = (v[m] - tau) * (v[m + 1] - tau);
        if (p < 0.0f && m < best) best = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if (lane == 0) {
        const float v0 = v[0] - tau;
        const bool first_free = v0 < 0.0f;
        bool mask = false;
        float dl = 0.0f, dh = 1.0f, fl = -1.0f, fh = 1.0f;
        if (best < M) {
            const int i2 = best + 1 < M - 1 ? best + 1 : M - 1;
            const float f_lo = v[best] - tau;
            mask = f_lo < 0.0f && first_free;
            if (mask) {
                const float fr = far[ray];
                dl = near * omu[best] + fr * u[best];  // the sweep depth number i (rendering.py:447-453)
                dh = near * omu[i2] + fr * u[i2];
                fl = f_lo;
                fh = v[i2] - tau;
            }
        }
        bracket[ray] = dl; bracket[n + ray] = dh; bracket[2 * n + ray] = fl; bracket[3 * n + ray] = fh;
        flags[ray] = (mask ? 1 : 0) | (first_free ? 2 : 0);
    }
}
}  // namespace psn

extern "C" int psn_first_crossing(const float* occ, const float* far, const float* u, const float* omu, float near, float tau,
                                  int64_t n_rays, int n_steps, float* bracket, int* flags, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(occ && far && u && omu && bracket && flags, "first_crossing: null pointer");
    PSN_CHECK_ARG(n_steps >= 2, "first_crossing: n_steps=%d", n_steps);
    if (n_rays <= 0) return PSN_OK;
    hipLaunchKernelGGL(first_crossing_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, occ, far, u, omu,
                       near, tau, n_rays, n_steps, bracket, flags);
    PSN_CHECK_LAUNCH("first_crossing");
    return PSN_OK;
}

// ---- shadow-ray sample points inside the object box (stage1/model/rendering.py:378-408) ------------------------------
// light_visibility evaluates the occupancy network on n_steps points of every (light, surface point) ray and then sets
// the occupancy of the points outside the [-box, box]^3 cube to zero.  Those points need no network evaluation at all:
// this kernel generates p = surf[s] + ldir[l] * d[m] (the reference's op order), tests the cube, and COMPACTS the rows
// that are inside (wave ballot + one atomic per wave) into pts [k, 3] with their dense row number rows[k] = (l ns + s)
// n_steps + m.  For an object of radius ~0.6 in the 1.1 cube about a quarter of the 128 samples of a ray survive, so
// the shadow-ray pass of shape_extract does ~4x fewer network rows with bit-identical visibility (the rows left out
// contribute alpha = 0 to the composite exactly as in the reference).  Order of the compacted rows is not deterministic;
// the value of every row is.
namespace psn {
__global__ __launch_bounds__(256) void shadow_points_kernel(const float* __restrict__ surf, const float* __restrict__ ldir,
                                                            int64_t ns, int nl, int S, float lnear, float lfar,
                                                            const float* __restrict__ u, const float* __restrict__ omu, float box,
                                                            float* __restrict__ pts, int64_t* __restrict__ rows,
                                                            unsigned long long* __restrict__ counter) {
    const int64_t total = (int64_t)nl * ns * S;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool inside = false;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (e < total) {
        const int m = (int)(e % S);
        const int64_t ray = e / S;
        const int64_t sidx = ray % ns;
        const int l = (int)(ray / ns);
        const float d = lnear * omu[m] + lfar * u[m];
        px = surf[sidx * 3 + 0] + ldir[l * 3 + 0] * d;
        py = surf[sidx * 3 + 1] + ldir[l * 3 + 1] * d;
        pz = surf[sidx * 3 + 2] + ldir[l * 3 + 2] * d;
        inside = px <= box && py <= box && pz <= box && px >= -box && py >= -box && pz >= -box;
    }
    const unsigned long long bal = __ballot(inside);
    const int lane = threadIdx.x & 63;
    const int cnt = __popcll(bal);
    unsigned long long base = 0;
    if (lane == 0 && cnt > 0) base = atomicAdd(counter, (unsigned long long)cnt);
    base = __shfl(base, 0);
    if (inside) {
        const unsigned long long k = base + __popcll(bal & ((1ull << lane) - 1ull));
        pts[k * 3 + 0] = px; pts[k * 3 + 1] = py; pts[k * 3 + 2] = pz;
        rows[k] = e;
    }
}
}  // namespace psn

extern "C" int psn_shadow_points(const float* surf, const float* ldir, int64_t n_surf, int n_lights, int n_steps, float lnear,
                                 float lfar, const float* u, const float* omu, float box, float* pts, int64_t* rows,
                                 unsigned long long* counter, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(surf && ldir && u && omu && pts && rows && counter, "shadow_points: null pointer");
    PSN_CHECK_ARG(n_steps >= 1 && n_lights >= 0 && n_surf >= 0, "shadow_points: bad sizes");
    const int64_t total = (int64_t)n_lights * n_surf * n_steps;
    if (total <= 0) return PSN_OK;
    const int64_t blocks = (total + 255) / 256;
    PSN_CHECK_ARG(blocks < (1ll << 31), "shadow_points: too many samples per call");
    hipLaunchKernelGGL(shadow_points_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, surf, ldir, n_surf, n_lights,
                       n_steps, lnear, lfar, u, omu, box, pts, rows, counter);
    PSN_CHECK_LAUNCH("shadow_points");
    return PSN_OK;
}
