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
    // the index list goes through LDS in chunks of 256: read straight from memory, every one of the n_idx iterations of every
    // thread waited for its own (dependent, branch-guarded) load -- 44 us for 96 lights, at the very end of a step's backward
    __shared__ int64_t sidx[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ai = 0.f;
    for (int i0 = 0; i0 < n_idx; i0 += 256) {
        const int m = n_idx - i0 < 256 ? n_idx - i0 : 256;
        __syncthreads();
        if ((int)threadIdx.x < m) sidx[threadIdx.x] = idx[i0 + threadIdx.x];
        __syncthreads();
        if (r < n_rows) {
            for (int j = 0; j < m; ++j) {
                if (sidx[j] != r) continue;
                const int i = i0 + j;
                if (g_dir != nullptr) {
                    float d0, d1, d2;
                    normalize3_bwd(dir_tab[3 * r], dir_tab[3 * r + 1], dir_tab[3 * r + 2], g_dir[3 * i], g_dir[3 * i + 1], g_dir[3 * i + 2], eps, d0, d1, d2);
                    a0 += d0; a1 += d1; a2 += d2;
                }
                if (g_int != nullptr) ai += g_int[i];
            }
        }
    }
    if (r >= n_rows) return;
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
struct AdamArgs { float* p; const float* g; float* m; float* v; float w1, beta2, w2, eps; AdamSeg seg[PSN_ADAM_MAX_SEGS]; int n_seg;
                  const float* dev; };  // dev: nullptr, or [n_seg][2] floats ON THE DEVICE that replace (neg_step, bc2_sqrt) of every range
__global__ __launch_bounds__(256) void adam_flat_kernel(AdamArgs a) {
    AdamSeg s = a.seg[blockIdx.y];
    if (a.dev != nullptr) { s.neg_step = a.dev[2 * blockIdx.y]; s.bc2_sqrt = a.dev[2 * blockIdx.y + 1]; }
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

// ---- step-front helpers: each replaces three or four launch-bound torch kernels (8 - 10 us apiece inside a replayed graph) ----
namespace psn {
// out[0] = number of i with a[i] && (b == nullptr || b[i]) as a float (exact below 2^24): bitwise_and + sum + two casts in torch
__global__ __launch_bounds__(1024) void mask_count_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, int64_t n,
                                                          float* __restrict__ out) {
    __shared__ int part[16];
    int c = 0;
    int64_t done = 0;
    if (((((uintptr_t)a) | ((uintptr_t)b)) & 7) == 0) {
        // eight mask bytes per load (one byte per load and 32 dependent-latency iterations per thread took 28 us at 32768 pixels);
        // high bit of every NON-ZERO byte: ((x & 0x7f..) + 0x7f..) | x
        const unsigned long long lo7 = 0x7f7f7f7f7f7f7f7full, hi1 = 0x8080808080808080ull;
        const int64_t n8 = n >> 3;
        for (int64_t i = threadIdx.x; i < n8; i += 1024) {
            const unsigned long long x = reinterpret_cast<const unsigned long long*>(a)[i];
            unsigned long long m = (((x & lo7) + lo7) | x) & hi1;
            if (b != nullptr) {
                const unsigned long long y = reinterpret_cast<const unsigned long long*>(b)[i];
                m &= (((y & lo7) + lo7) | y);
            }
            c += __popcll(m);
        }
        done = n8 << 3;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += 1024) c += (a[i] != 0 && (b == nullptr || b[i] != 0)) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = (float)t;
    }
}
// inv[p] = the position of pixel p in the ASCENDING list idx[0 .. ns), or -1: fill + arange + index_put in torch
__global__ __launch_bounds__(256) void inverse_index_kernel(const int64_t* __restrict__ idx, int64_t ns, int64_t n_pix, int* __restrict__ inv,
                                                            const float* __restrict__ count) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    if (count != nullptr) {  // a list padded to a fixed length: only its first count[0] entries are real (0: no pixel has a row)
        const int64_t c = (int64_t)count[0];
        ns = c < ns ? c : ns;
    }
    int64_t lo = 0, hi = ns;  // first position with idx[pos] >= p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (idx[mid] < p) lo = mid + 1; else hi = mid;
    }
    inv[p] = (lo < ns && idx[lo] == p) ? (int)lo : -1;
}
}  // namespace psn

// ---- up to PSN_COPY2D_MAX strided 2-D copies in one launch (weight-pack side tables: init-table slices, bias segments) ----------
namespace psn {
struct Copy2dArgs { PsnCopy2dItem it[PSN_COPY2D_MAX]; int64_t start[PSN_COPY2D_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void copy2d_group_kernel(Copy2dArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int64_t)blockIdx.x >= a.start[i + 1]) ++i;
    const PsnCopy2dItem it = a.it[i];
    const int64_t e = ((int64_t)blockIdx.x - a.start[i]) * 256 + threadIdx.x;
    if (e >= (int64_t)it.rows * it.cols) return;
    const int64_t r = e / it.cols, c = e - r * it.cols;
    it.dst[r * it.ld_dst + c] = it.src[r * it.ld_src + c];
}
}  // namespace psn

extern "C" int psn_copy2d_group(int n_items, const PsnCopy2dItem* items, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= PSN_COPY2D_MAX, "copy2d_group: n_items=%d", n_items);
    Copy2dArgs a;
    a.n = n_items;
    a.start[0] = 0;
    for (int i = 0; i < n_items; ++i) {
        const PsnCopy2dItem& it = items[i];
        PSN_CHECK_ARG(it.src && it.dst && it.rows >= 1 && it.cols >= 1 && it.ld_src >= it.cols && it.ld_dst >= it.cols, "copy2d_group: item %d", i);
        a.it[i] = it;
        a.start[i + 1] = a.start[i] + ((int64_t)it.rows * it.cols + 255) / 256;
    }
    hipLaunchKernelGGL(copy2d_group_kernel, dim3((unsigned)a.start[n_items]), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("copy2d_group");
    return PSN_OK;
}

// ---- up to PSN_COPY_BYTES_MAX contiguous copies of any element type in one launch (a batch -> the input buffers of a HIP graph) ----
namespace psn {
struct CopyBytesArgs { PsnCopyBytesItem it[PSN_COPY_BYTES_MAX]; int64_t start[PSN_COPY_BYTES_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void copy_bytes_group_kernel(CopyBytesArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int64_t)blockIdx.x >= a.start[i + 1]) ++i;
    const PsnCopyBytesItem it = a.it[i];
    const int64_t b = (int64_t)blockIdx.x - a.start[i];  // block of 256 threads x 16 bytes
    const int64_t off = (b * 256 + threadIdx.x) * 16;
    if (off >= it.n_bytes) return;
    const char* s = reinterpret_cast<const char*>(it.src) + off;
    char* d = reinterpret_cast<char*>(it.dst) + off;
    if (it.aligned && off + 16 <= it.n_bytes) {
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    } else {
        const int m = (int)(it.n_bytes - off < 16 ? it.n_bytes - off : 16);
        for (int k = 0; k < m; ++k) d[k] = s[k];
    }
}
}  // namespace psn

extern "C" int psn_copy_bytes_group(int n_items, const PsnCopyBytesItem* items, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= PSN_COPY_BYTES_MAX, "copy_bytes_group: n_items=%d", n_items);
    CopyBytesArgs a;
    a.n = n_items;
    a.start[0] = 0;
    for (int i = 0; i < n_items; ++i) {
        PSN_CHECK_ARG(items[i].src && items[i].dst && items[i].n_bytes >= 1, "copy_bytes_group: item %d", i);
        a.it[i] = items[i];
        a.it[i].aligned = ((((uintptr_t)items[i].src) | ((uintptr_t)items[i].dst)) & 15) == 0;
        a.start[i + 1] = a.start[i] + (items[i].n_bytes + 4095) / 4096;
    }
    PSN_CHECK_ARG(a.start[n_items] < (1ll << 31), "copy_bytes_group: too many bytes");
    hipLaunchKernelGGL(copy_bytes_group_kernel, dim3((unsigned)a.start[n_items]), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("copy_bytes_group");
    return PSN_OK;
}

// idx[0 .. ns) = the positions of the set bytes of mask [n] in ascending order, idx[ns .. cap) = idx[ns - 1] (0 when the mask is
// empty), count[0] = ns as a float: nonzero() without its host synchronisation, into a FIXED-size list (graph replay).
namespace psn {
__global__ __launch_bounds__(1024) void surface_index_kernel(const unsigned char* __restrict__ mask, int64_t n, int64_t cap, int64_t* __restrict__ idx,
                                                             float* __restrict__ count) {
    __shared__ int cnt[1024];
    __shared__ int total, last;
    const int t = threadIdx.x;
    const int64_t q = (n + 1023) / 1024, lo = t * q, hi = lo + q < n ? lo + q : n;
    int c = 0;
    for (int64_t i = lo; i < hi; ++i) c += mask[i] != 0;
    cnt[t] = c;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < 1024; ++k) { const int v = cnt[k]; cnt[k] = run; run += v; }
        total = run;
        last = 0;
    }
    __syncthreads();
    int64_t pos = cnt[t];
    for (int64_t i = lo; i < hi; ++i)
        if (mask[i] != 0) {
            if (pos < cap) idx[pos] = i;
            if (pos == (int64_t)total - 1) last = (int)i;
            ++pos;
        }
    __syncthreads();
    const int ns = total < cap ? total : (int)cap;
    for (int64_t k = ns + t; k < cap; k += 1024) idx[k] = last;
    if (t == 0 && count != nullptr) count[0] = (float)total;
}
}  // namespace psn

extern "C" int psn_surface_index(const unsigned char* mask, int64_t n, int64_t cap, int64_t* idx, float* count, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(mask && idx && n >= 0 && n < (1ll << 24) && cap >= 1, "surface_index: bad arguments (n=%lld cap=%lld)", (long long)n, (long long)cap);
    hipLaunchKernelGGL(surface_index_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, n, cap, idx, count);
    PSN_CHECK_LAUNCH("surface_index");
    return PSN_OK;
}

extern "C" int psn_mask_count(const unsigned char* mask_a, const unsigned char* mask_b, int64_t n, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(mask_a && out && n >= 0 && n < (1ll << 24), "mask_count: bad arguments (n=%lld: the count must be exact in fp32)", (long long)n);
    hipLaunchKernelGGL(mask_count_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask_a, mask_b, n, out);
    PSN_CHECK_LAUNCH("mask_count");
    return PSN_OK;
}

extern "C" int psn_inverse_index(const int64_t* idx, int64_t ns, int64_t n_pix, int* inv, const float* count, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG((idx || ns == 0) && inv && ns >= 0 && n_pix >= 0 && ns < (1ll << 31), "inverse_index: bad arguments");
    if (n_pix == 0) return PSN_OK;
    hipLaunchKernelGGL(inverse_index_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, ns, n_pix, inv, count);
    PSN_CHECK_LAUNCH("inverse_index");
    return PSN_OK;
}

static int adam_flat_impl(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                          float one_minus_beta1, float beta2, float one_minus_beta2, float eps, const float* seg_scalars_dev, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && segs && n_segs >= 1 && n_segs <= PSN_ADAM_MAX_SEGS, "adam_flat: bad arguments (n_segs=%d)", n_segs);
    AdamArgs a = {};
    a.dev = seg_scalars_dev;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.w1 = one_minus_beta1; a.beta2 = beta2; a.w2 = one_minus_beta2; a.eps = eps;
    a.n_seg = n_segs;
    int64_t max_n = 0;
    for (int i = 0; i < n_segs; ++i) {
        PSN_CHECK_ARG(segs[i].offset >= 0 && segs[i].grad_offset >= 0 && segs[i].n >= 0 && (seg_scalars_dev != nullptr || segs[i].bias_correction2_sqrt > 0.0f),
                      "adam_flat: segment %d", i);
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

extern "C" int psn_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                             float one_minus_beta1, float beta2, float one_minus_beta2, float eps, void* stream) {
    return adam_flat_impl(param, grad, exp_avg, exp_avg_sq, n_segs, segs, one_minus_beta1, beta2, one_minus_beta2, eps, nullptr, stream);
}

// The same update with the step-dependent scalars of every range read from DEVICE memory (seg_scalars_dev [n_segs][2] =
// neg_step_size, bias_correction2_sqrt): the launch is then identical from step to step and can be replayed from a HIP graph
// while the host refreshes the scalars (stage2/graph.py).
extern "C" int psn_adam_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                                 float one_minus_beta1, float beta2, float one_minus_beta2, float eps, const float* seg_scalars_dev,
                                 void* stream) {
    PSN_CHECK_ARG(seg_scalars_dev != nullptr, "adam_flat_dev: seg_scalars_dev is required");
    return adam_flat_impl(param, grad, exp_avg, exp_avg_sq, n_segs, segs, one_minus_beta1, beta2, one_minus_beta2, eps, seg_scalars_dev, stream);
}

// ---- stage-1 per-ray glue (round 3): the ~100 elementwise launches on [N] / [N, 3] tensors around the ray march ------------
//   psn_stage1_rays      common.py:205-226 + rendering.py:576-596: camera origin, normalised ray directions, sphere exit depth
//   psn_surface_points   rendering.py:516-522 + :84-108: d_i from (root-finder depth, crossing flags), masks, surface points
//   psn_stage1_targets   common.py:172-202 x 5 + training.py:176-191: nearest-pixel ground truth of the sampled pixels
// Arithmetic in the op order of the torch formulations they replace (products and sums rounded separately): same bits.
namespace psn {

__global__ __launch_bounds__(256) void stage1_rays_kernel(const float* __restrict__ pix, const float* __restrict__ K, int k_ld,
                                                          const float* __restrict__ W, float radius2, int64_t n, float* __restrict__ cam, float* __restrict__ rays,
                                                          float* __restrict__ far) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float fx = K[0], cx = K[2], cy = K[k_ld + 2];
    const float qx = (pix[2 * i] - cx) / fx, qy = (pix[2 * i + 1] - cy) / fx;   // both axes over fx: the reference's quirk (common.py:220)
    float d[3], c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] = qx * W[4 * r] + qy * W[4 * r + 1] + W[4 * r + 2];
        c[r] = W[4 * r + 3];
    }
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { d[r] = d[r] / nrm; rays[3 * i + r] = d[r]; cam[3 * i + r] = c[r]; }
    // rendering.py:576-596: b = ray . cam, under = b^2 - (|cam|^2 - r^2), far = max(sqrt(under) - b, 0) where under > 0
    const float b = d[0] * c[0] + d[1] * c[1] + d[2] * c[2];
    const float cn = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float under = b * b - (cn * cn - radius2);
    float f = 0.0f;
    if (under > 0.0f) {
        f = sqrtf(under) - b;
        f = f < 0.0f ? 0.0f : f;
    }
    far[i] = f;
}

__global__ __launch_bounds__(256) void surface_points_kernel(const float* __restrict__ d_pred, const int* __restrict__ flags,
                                                             const float* __restrict__ cam, const float* __restrict__ rays, int64_t n,
                                                             float* __restrict__ d_i, float* __restrict__ dists,
                                                             unsigned char* __restrict__ obj_mask, float* __restrict__ points) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int fl = flags[i];
    float d = (fl & 1) ? d_pred[i] : __builtin_inff();   // rendering.py:519-521
    d = (fl & 2) ? d : 0.0f;                             // :522
    const bool zero_occ = d == 0.0f;
    const bool ok = !(fabsf(d) == __builtin_inff()) && !(d != d);
    float dist = ok ? d : 1.0f;
    dist = zero_occ ? 0.0f : dist;
    if (d_i != nullptr) d_i[i] = d;
    dists[i] = dist;
    obj_mask[i] = (ok && !zero_occ) ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) points[3 * i + c] = cam[3 * i + c] + rays[3 * i + c] * dist;
}

// nearest-neighbour grid_sample(align_corners=True, zeros padding) at x = 2 px / w - 1, y = 2 py / h - 1 (common.py:190-195):
// ATen: ix = ((x + 1) / 2) * (W - 1), nearbyint, in-bounds or 0
__device__ __forceinline__ bool nearest_index(float px, float py, int h, int w, int& iy, int& ix) {
    // tensor / python scalar on the GPU is a multiplication by the fp32 reciprocal (ATen div_true_kernel_cuda)
    const float x = 2.0f * px * (1.0f / (float)w) - 1.0f, y = 2.0f * py * (1.0f / (float)h) - 1.0f;
    const float fx = ((x + 1.0f) / 2.0f) * (float)(w - 1), fy = ((y + 1.0f) / 2.0f) * (float)(h - 1);
    const float rx = rintf(fx), ry = rintf(fy);
    ix = (int)rx; iy = (int)ry;
    return rx >= 0.0f && rx <= (float)(w - 1) && ry >= 0.0f && ry <= (float)(h - 1);
}
struct TargetArgs {
    const float* pix; const float* img; const float* mask; const float* mask_valid; const float* normal; const float* norm_mask;
    const float* W;   // world_mat [4, 4] (rotation of the normal ground truth, training.py:191) or nullptr
    float cos_thresh; int use_angle; int h, w; int64_t n;
    float* rgb_gt; float* mask_gt; unsigned char* mask_valid_out; float* normal_gt; unsigned char* norm_mask_out;
};
__global__ __launch_bounds__(256) void stage1_targets_kernel(TargetArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    int iy, ix;
    const bool in = nearest_index(a.pix[2 * i], a.pix[2 * i + 1], a.h, a.w, iy, ix);
    const int64_t hw = (int64_t)a.h * a.w, o = in ? (int64_t)iy * a.w + ix : 0;
    auto at = [&](const float* p, int c) -> float { return in ? p[c * hw + o] : 0.0f; };
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb_gt[3 * i + c] = at(a.img, c);
    if (a.mask_gt != nullptr) a.mask_gt[i] = (a.mask != nullptr ? at(a.mask, 0) : 1.0f) != 0.0f ? 1.0f : 0.0f;
    if (a.mask_valid_out != nullptr) a.mask_valid_out[i] = (a.mask_valid != nullptr ? at(a.mask_valid, 0) : 1.0f) != 0.0f ? 1 : 0;
    bool nm = a.norm_mask != nullptr ? at(a.norm_mask, 0) != 0.0f : false;
    if (a.normal_gt != nullptr) {
        const float n0 = at(a.normal, 0), n1 = at(a.normal, 1), n2 = at(a.normal, 2);
        if (a.use_angle && n2 < a.cos_thresh) nm = false;   // training.py:189-190: on the UNROTATED normal
        // training.py:191: R diag(1, -1, -1) n, the products of the broadcast formulation summed left to right
#pragma unroll
        for (int r = 0; r < 3; ++r)
            a.normal_gt[3 * i + r] = n0 * (a.W[4 * r] * 1.0f) + n1 * (a.W[4 * r + 1] * -1.0f) + n2 * (a.W[4 * r + 2] * -1.0f);
    }
    if (a.norm_mask_out != nullptr) a.norm_mask_out[i] = nm ? 1 : 0;
}

}  // namespace psn

extern "C" int psn_stage1_rays(const float* pix, const float* camera_mat, int k_ld, const float* world_mat, float radius2, int64_t n,
                               float* cam, float* rays, float* far, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(pix && camera_mat && world_mat && cam && rays && far && n >= 0, "stage1_rays: null pointer");
    PSN_CHECK_ARG(k_ld == 3 || k_ld == 4, "stage1_rays: camera_mat is 3 x 3 or 4 x 4, row-major");
    if (n == 0) return PSN_OK;
    hipLaunchKernelGGL(stage1_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pix, camera_mat, k_ld,
                       world_mat, radius2, n, cam, rays, far);
    PSN_CHECK_LAUNCH("stage1_rays");
    return PSN_OK;
}

extern "C" int psn_surface_points(const float* d_pred, const int* flags, const float* cam, const float* rays, int64_t n, float* d_i,
                                  float* dists, unsigned char* obj_mask, float* points, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(d_pred && flags && cam && rays && dists && obj_mask && points && n >= 0, "surface_points: null pointer");
    if (n == 0) return PSN_OK;
    hipLaunchKernelGGL(surface_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_pred, flags, cam, rays, n,
                       d_i, dists, obj_mask, points);
    PSN_CHECK_LAUNCH("surface_points");
    return PSN_OK;
}

extern "C" int psn_stage1_targets(const float* pix, int64_t n, int h, int w, const float* img, const float* mask, const float* mask_valid,
                                  const float* normal, const float* norm_mask, const float* world_mat, int use_angle, float cos_thresh,
                                  float* rgb_gt, float* mask_gt, unsigned char* mask_valid_out, float* normal_gt,
                                  unsigned char* norm_mask_out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(pix && img && rgb_gt && n >= 0 && h >= 1 && w >= 1, "stage1_targets: null pointer / empty image");
    PSN_CHECK_ARG(normal_gt == nullptr || (normal && world_mat), "stage1_targets: the normal ground truth needs the normal image and world_mat");
    if (n == 0) return PSN_OK;
    TargetArgs a = {};
    a.pix = pix; a.img = img; a.mask = mask; a.mask_valid = mask_valid; a.normal = normal; a.norm_mask = norm_mask; a.W = world_mat;
    a.cos_thresh = cos_thresh; a.use_angle = use_angle; a.h = h; a.w = w; a.n = n;
    a.rgb_gt = rgb_gt; a.mask_gt = mask_gt; a.mask_valid_out = mask_valid_out; a.normal_gt = normal_gt; a.norm_mask_out = norm_mask_out;
    hipLaunchKernelGGL(stage1_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("stage1_targets");
    return PSN_OK;
}
