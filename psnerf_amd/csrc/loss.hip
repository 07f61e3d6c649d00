// Stage-2 losses in two launches forward and one backward (stage2/model/loss.py:27-58,76-92,123-141):
//   rgb:      mean over masked pixels, lights, channels of |rgb - gt| (L1) or (rgb - gt)^2 (L2)
//   albedo / SG-weight smoothness: mean |x - x_jitter|          visibility: mean |vis_train[..., 0] - vis_gt| (or ^2)
//   normal:   mean (n_pred - normalize(n_stage1))^2             normal smoothness: mean |n_pred - n_jitter|
// with mask = network_object_mask & object_mask and count = number of masked pixels (summed over ranks by the caller).
// The torch formulation is ~35 elementwise / reduction launches forward and ~40 backward on [L, N, 3] tensors that are
// each a few MB -- all latency-bound; here every dense tensor is read once and every gradient written once.
// HBM-bound: forward reads 24 L + ~300 B per pixel, backward writes 12 L + ~150 B per pixel.
#include "common.h"

namespace psn {

struct LossArgs {
    const float* rgb; const float* rgb_gt; int L;          // [L, N, 3]
    const float* alb; const float* alb_j;                  // [N, 3] or null
    const float* wgt; const float* wgt_j; int nb;          // [N, nb] or null
    const float* vis; const float* vis_gt; int V;          // vis [V, N, 3] (channel 0), gt [V, N]; or null
    const float* nrm; const float* nrm_gt; const float* nrm_j;  // [N, 3]; nrm_gt is normalised here; nrm / nrm_j may be null
    const unsigned char* mask_a; const unsigned char* mask_b;   // [N] bool each
    int64_t N;
    int l2;                                                // image / visibility loss: 0 = L1, 1 = L2
    float* partial;                                        // [blocks, 6]
    // backward
    const float* g_total;                                  // [1] upstream gradient of the weighted total
    const float* count_dev;                                // optional [1]: every k_* is divided by it (see LossScale)
    float k_rgb, k_alb, k_wgt, k_vis, k_nrm, k_nrmj;       // weight_i / denominator_i
    float* d_rgb; float* d_alb; float* d_alb_j; float* d_wgt; float* d_wgt_j; float* d_vis; float* d_nrm; float* d_nrm_j;
};

__device__ __forceinline__ float img_term(float a, float b, int l2) { const float d = a - b; return l2 ? d * d : fabsf(d); }
__device__ __forceinline__ float img_grad(float a, float b, int l2) {
    const float d = a - b;
    return l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
}

__global__ __launch_bounds__(256) void stage2_loss_fwd_kernel(LossArgs a) {
    // blockIdx.y selects a chunk of lights for the [L, N, 3] term (enough independent loads in flight to stream it);
    // the per-pixel terms are computed by chunk 0 only
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lc = (a.L + gridDim.y - 1) / gridDim.y;
    const int l0 = blockIdx.y * lc, l1 = min(a.L, l0 + lc);
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.x * 256) {
        if (!(a.mask_a[n] && a.mask_b[n])) continue;
        if (a.rgb != nullptr) {
#pragma unroll 4
            for (int l = l0; l < l1; ++l) {
                const float* p = a.rgb + ((int64_t)l * a.N + n) * 3;
                const float* q = a.rgb_gt + ((int64_t)l * a.N + n) * 3;
                s[0] += img_term(p[0], q[0], a.l2) + img_term(p[1], q[1], a.l2) + img_term(p[2], q[2], a.l2);
            }
        }
        if (blockIdx.y != 0) continue;
        if (a.alb != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s[1] += fabsf(a.alb[n * 3 + c] - a.alb_j[n * 3 + c]);
        }
        if (a.wgt != nullptr) {
            for (int k = 0; k < a.nb; ++k) s[2] += fabsf(a.wgt[n * a.nb + k] - a.wgt_j[n * a.nb + k]);
        }
        if (a.vis != nullptr) {
            for (int v = 0; v < a.V; ++v) s[3] += img_term(a.vis[((int64_t)v * a.N + n) * 3], a.vis_gt[(int64_t)v * a.N + n], a.l2);
        }
        if (a.nrm != nullptr) {
            const float gx = a.nrm_gt[n * 3], gy = a.nrm_gt[n * 3 + 1], gz = a.nrm_gt[n * 3 + 2];
            const float inv = 1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize(dim=-1)
            const float dx = a.nrm[n * 3] - gx * inv, dy = a.nrm[n * 3 + 1] - gy * inv, dz = a.nrm[n * 3 + 2] - gz * inv;
            s[4] += dx * dx + dy * dy + dz * dz;
            if (a.nrm_j != nullptr) {
#pragma unroll
                for (int c = 0; c < 3; ++c) s[5] += fabsf(a.nrm[n * 3 + c] - a.nrm_j[n * 3 + c]);
            }
        }
    }
    __shared__ float red[4][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v = s[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
        a.partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[0..5] = term_i = sum_i / denom_i, out[6] = sum_i weight_i term_i   (fixed summation order: deterministic)
// inv_denom[i] is the reciprocal of term i's element count; with count_dev != null it is only the per-pixel multiplicity
// (1 / (L 3), 1 / 3, ...) and the masked-pixel count is read from the device (a count that was all-reduced over ranks, or
// simply never brought to the host: no synchronisation between data loading and the first launch of the step).
struct LossScale { float inv_denom[6]; float weight[6]; const float* count_dev; };
// 256 threads: thread t sums slice t >> 3 (of 32) of term t & 7; fixed tree afterwards: deterministic
__global__ __launch_bounds__(256) void stage2_loss_final_kernel(const float* __restrict__ partial, int blocks, LossScale sc,
                                                               float* __restrict__ out) {
    __shared__ float red[32][8];
    const int i = threadIdx.x & 7, j = threadIdx.x >> 3;
    float s = 0.f;
    if (i < 6)
        for (int b = j; b < blocks; b += 32) s += partial[(int64_t)b * 6 + i];
    red[j][i] = s;
    __syncthreads();
    if (threadIdx.x < 8) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += red[k][threadIdx.x];
        const float cnt = sc.count_dev != nullptr ? sc.count_dev[0] : 1.0f;
        const float term = (threadIdx.x < 6 && cnt > 0.0f) ? tot * sc.inv_denom[threadIdx.x] / cnt : 0.f;
        if (threadIdx.x < 6) out[threadIdx.x] = term;
        float w = threadIdx.x < 6 ? sc.weight[threadIdx.x] * term : 0.f;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
        if (threadIdx.x == 0) out[6] = w;
    }
}

// Sums of a [V, Ns, C] tensor over its first and over its second dimension in ONE pass (C <= 256 columns = threads):
//   sx[n, c] = sum_v x[v, n, c]        sl_part[chunk, v, c] = sum over the chunk's n of x[v, n, c]
// (separable input-block weight gradient of the visibility network, ops.VisibilityPair.backward)
__global__ __launch_bounds__(256) void pair_sums_kernel(const float* __restrict__ x, int V, int64_t Ns, int C, int rows_per_block,
                                                        float* __restrict__ sx, float* __restrict__ sl_part) {
    // thread = (column quad cg, row lane rl): 16-byte loads, four rows of the block's slab in flight per light
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_block, n1 = min(Ns, n0 + rows_per_block);
    const bool live = 4 * cg < C;  // C is a multiple of 4
    float4 sl[PSN_PAIR_SUMS_MAX_V];
#pragma unroll
    for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v) sl[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int64_t n = n0 + rl; n < n1; n += 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v) {
                if (v < V) {
                    const float4 t = *reinterpret_cast<const float4*>(x + ((int64_t)v * Ns + n) * C + 4 * cg);
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                    sl[v].x += t.x; sl[v].y += t.y; sl[v].z += t.z; sl[v].w += t.w;
                }
            }
            *reinterpret_cast<float4*>(sx + n * C + 4 * cg) = acc;
        }
#pragma unroll
        for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v)
            if (v < V) *reinterpret_cast<float4*>(sl_part + (((int64_t)blockIdx.x * 4 + rl) * V + v) * C + 4 * cg) = sl[v];
    }
}

// ---- the separable input-block weight gradients of ops.VisibilityPair.backward for ALL input layers in two launches -----------
// (were, per layer: pair_sums + a torch reduction of its partials + a [C, V] x [V, 64] GEMM, + a bias reduction: 7 launches)
struct PairGroupArgs { PsnPairSumsItem it[PSN_PAIR_GROUP_MAX]; int V; int64_t Ns; int C; int rows_per_block; int chunks; float* part;
                       const float* pe_l; int64_t ld_pe; int n_pe; int64_t ld_w;
                       int slices; float* part2; };  // slices > 1: kernel B runs twice (chunk slices -> part2, then part2 -> results)
// A: as pair_sums_kernel for item blockIdx.y, the four row lanes of a block combined through LDS: part [item][chunk][V][C]
__global__ __launch_bounds__(256) void pair_group_sums_kernel(PairGroupArgs a) {
    const PsnPairSumsItem it = a.it[blockIdx.y];
    const int V = a.V, C = a.C;
    const int64_t Ns = a.Ns;
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * a.rows_per_block, n1 = min(Ns, n0 + a.rows_per_block);
    const bool live = 4 * cg < C;
    float4 sl[PSN_PAIR_SUMS_MAX_V];
#pragma unroll
    for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v) sl[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int64_t n = n0 + rl; n < n1; n += 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v) {
                if (v < V) {
                    const float4 t = *reinterpret_cast<const float4*>(it.x + ((int64_t)v * Ns + n) * C + 4 * cg);
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                    sl[v].x += t.x; sl[v].y += t.y; sl[v].z += t.z; sl[v].w += t.w;
                }
            }
            *reinterpret_cast<float4*>(it.sx + n * C + 4 * cg) = acc;
        }
    }
    __shared__ float4 red[3][64];
    float* dst = a.part + (((int64_t)blockIdx.y * a.chunks + blockIdx.x) * V) * C;
#pragma unroll
    for (int v = 0; v < PSN_PAIR_SUMS_MAX_V; ++v) {
        if (v < V) {
            __syncthreads();
            if (rl > 0) red[rl - 1][cg] = sl[v];
            __syncthreads();
            if (rl == 0 && live) {
                float4 t = sl[v];
#pragma unroll
                for (int q = 0; q < 3; ++q) { const float4 o = red[q][cg]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }  // lanes 0 + 1 + 2 + 3: fixed order
                *reinterpret_cast<float4*>(dst + (int64_t)v * C + 4 * cg) = t;
            }
        }
    }
}
// B: block (item, 64-column group), thread (column, v slot): dz_l[v][c] = sum over chunks of part (fixed order), then
//    dWl[c][k] = sum_v dz_l[v][c] pe_l[v][k] (k < n_pe) and bias[c] = sum_v dz_l[v][c] for the block's 64 columns
__global__ __launch_bounds__(1024) void pair_group_final_kernel(PairGroupArgs a) {
    const PsnPairSumsItem it = a.it[blockIdx.y];
    const int V = a.V, C = a.C;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 63, v = threadIdx.x >> 6;  // v < 16
    __shared__ float dzl[PSN_PAIR_SUMS_MAX_V][64];
    __shared__ float pel[PSN_PAIR_SUMS_MAX_V][64];
    // the chunk range of this block: everything, or -- first pass of the sliced form -- slice blockIdx.z of it.  (With one block
    // per (item, column group) a thread walked ALL chunks: 1843 dependent 4-byte loads at Ns = 29487 -- 268 us on 8 blocks.)
    int ch = 0, ch_end = a.chunks;
    if (a.slices > 1) {
        const int per = (a.chunks + a.slices - 1) / a.slices;
        ch = blockIdx.z * per;
        ch_end = min(a.chunks, ch + per);
    }
    float s = 0.0f;
    if (v < V && c0 + cl < C) {
        const float* p = a.part + ((int64_t)blockIdx.y * a.chunks * V + v) * C + c0 + cl;
        const int64_t step = (int64_t)V * C;
        for (; ch + 4 <= ch_end; ch += 4) {
            const float t0 = p[(int64_t)ch * step], t1 = p[(int64_t)(ch + 1) * step], t2 = p[(int64_t)(ch + 2) * step], t3 = p[(int64_t)(ch + 3) * step];
            s += t0; s += t1; s += t2; s += t3;
        }
        for (; ch < ch_end; ++ch) s += p[(int64_t)ch * step];
        if (a.slices > 1) a.part2[(((int64_t)blockIdx.y * a.slices + blockIdx.z) * V + v) * C + c0 + cl] = s;
    }
    if (a.slices > 1) return;
    dzl[v][cl] = s;
    pel[v][cl] = (v < V && cl < a.n_pe) ? a.pe_l[(int64_t)v * a.ld_pe + cl] : 0.0f;
    __syncthreads();
    // 64 columns x n_pe (<= 64) outputs over 1024 threads
    for (int e = threadIdx.x; e < 64 * a.n_pe; e += 1024) {
        const int c = e / a.n_pe, k = e - c * a.n_pe;
        if (c0 + c < C) {
            float w = 0.0f;
            for (int u = 0; u < V; ++u) w += dzl[u][c] * pel[u][k];
            it.dWl[(int64_t)(c0 + c) * a.ld_w + k] = w;
        }
    }
    if (it.bias != nullptr && threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
        float b = 0.0f;
        for (int u = 0; u < V; ++u) b += dzl[u][threadIdx.x];
        it.bias[c0 + threadIdx.x] = b;
    }
}

__global__ __launch_bounds__(256) void stage2_loss_bwd_kernel(LossArgs a) {
    float g = a.g_total[0];
    if (a.count_dev != nullptr) {
        const float cnt = a.count_dev[0];
        g = cnt > 0.0f ? g / cnt : 0.0f;
    }
    const int lc = (a.L + gridDim.y - 1) / gridDim.y;
    const int l0 = blockIdx.y * lc, l1 = min(a.L, l0 + lc);
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.x * 256) {
        const bool m = a.mask_a[n] && a.mask_b[n];
        if (a.d_rgb != nullptr) {
            const float k = g * a.k_rgb;
#pragma unroll 4
            for (int l = l0; l < l1; ++l) {
                const int64_t o = ((int64_t)l * a.N + n) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) a.d_rgb[o + c] = m ? k * img_grad(a.rgb[o + c], a.rgb_gt[o + c], a.l2) : 0.f;
            }
        }
        if (blockIdx.y != 0) continue;
        if (a.d_alb != nullptr) {
            const float k = g * a.k_alb;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = m ? k * img_grad(a.alb[n * 3 + c], a.alb_j[n * 3 + c], 0) : 0.f;
                a.d_alb[n * 3 + c] = v;
                a.d_alb_j[n * 3 + c] = -v;
            }
        }
        if (a.d_wgt != nullptr) {
            const float k = g * a.k_wgt;
            for (int q = 0; q < a.nb; ++q) {
                const float v = m ? k * img_grad(a.wgt[n * a.nb + q], a.wgt_j[n * a.nb + q], 0) : 0.f;
                a.d_wgt[n * a.nb + q] = v;
                a.d_wgt_j[n * a.nb + q] = -v;
            }
        }
        if (a.d_vis != nullptr) {
            const float k = g * a.k_vis;
            for (int v = 0; v < a.V; ++v) {
                const int64_t o = ((int64_t)v * a.N + n) * 3;
                a.d_vis[o] = m ? k * img_grad(a.vis[o], a.vis_gt[(int64_t)v * a.N + n], a.l2) : 0.f;
                a.d_vis[o + 1] = 0.f;
                a.d_vis[o + 2] = 0.f;
            }
        }
        if (a.d_nrm != nullptr) {
            float gn[3] = {0.f, 0.f, 0.f}, gj[3] = {0.f, 0.f, 0.f};
            if (m) {
                const float gx = a.nrm_gt[n * 3], gy = a.nrm_gt[n * 3 + 1], gz = a.nrm_gt[n * 3 + 2];
                const float inv = 1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
                const float t[3] = {gx * inv, gy * inv, gz * inv};
#pragma unroll
                for (int c = 0; c < 3; ++c) gn[c] = g * a.k_nrm * 2.0f * (a.nrm[n * 3 + c] - t[c]);
                if (a.nrm_j != nullptr) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = g * a.k_nrmj * img_grad(a.nrm[n * 3 + c], a.nrm_j[n * 3 + c], 0);
                        gn[c] += v;
                        gj[c] = -v;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.d_nrm[n * 3 + c] = gn[c];
                if (a.d_nrm_j != nullptr) a.d_nrm_j[n * 3 + c] = gj[c];
            }
        }
    }
}

constexpr int kLossBlocksX = 256, kLossChunksY = 8;  // partial: [kLossBlocksX * kLossChunksY, 6] floats at most

}  // namespace psn

extern "C" int psn_stage2_loss_fwd(const float* rgb, const float* rgb_gt, int L, const float* alb, const float* alb_j,
                                   const float* wgt, const float* wgt_j, int nb, const float* vis, const float* vis_gt, int V,
                                   const float* nrm, const float* nrm_gt, const float* nrm_j, const unsigned char* mask_a,
                                   const unsigned char* mask_b, int64_t N, int l2, const float* inv_denom, const float* weight,
                                   const float* count_dev, float* partial, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(mask_a && mask_b && partial && out && inv_denom && weight, "stage2_loss_fwd: null pointer");
    PSN_CHECK_ARG((rgb == nullptr || rgb_gt) && (alb == nullptr || alb_j) && (wgt == nullptr || (wgt_j && nb > 0)) &&
                  (vis == nullptr || vis_gt) && (nrm == nullptr || nrm_gt), "stage2_loss_fwd: incomplete term");
    LossArgs a = {};
    a.rgb = rgb; a.rgb_gt = rgb_gt; a.L = L; a.alb = alb; a.alb_j = alb_j; a.wgt = wgt; a.wgt_j = wgt_j; a.nb = nb;
    a.vis = vis; a.vis_gt = vis_gt; a.V = V; a.nrm = nrm; a.nrm_gt = nrm_gt; a.nrm_j = nrm_j; a.mask_a = mask_a; a.mask_b = mask_b;
    a.N = N; a.l2 = l2; a.partial = partial;
    int bx = (int)((N + 255) / 256);
    if (bx > kLossBlocksX) bx = kLossBlocksX;
    if (bx < 1) bx = 1;
    const int by = (rgb != nullptr && L >= 2 * kLossChunksY) ? kLossChunksY : 1;
    hipLaunchKernelGGL(stage2_loss_fwd_kernel, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("stage2_loss_fwd");
    LossScale sc;
    for (int i = 0; i < 6; ++i) { sc.inv_denom[i] = inv_denom[i]; sc.weight[i] = weight[i]; }
    sc.count_dev = count_dev;
    hipLaunchKernelGGL(stage2_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, bx * by, sc, out);
    PSN_CHECK_LAUNCH("stage2_loss_fwd (final)");
    return PSN_OK;
}

extern "C" int psn_stage2_loss_bwd(const float* g_total, const float* rgb, const float* rgb_gt, int L, float k_rgb, float* d_rgb,
                                   const float* alb, const float* alb_j, float k_alb, float* d_alb, float* d_alb_j,
                                   const float* wgt, const float* wgt_j, int nb, float k_wgt, float* d_wgt, float* d_wgt_j,
                                   const float* vis, const float* vis_gt, int V, float k_vis, float* d_vis,
                                   const float* nrm, const float* nrm_gt, const float* nrm_j, float k_nrm, float k_nrmj, float* d_nrm,
                                   float* d_nrm_j, const unsigned char* mask_a, const unsigned char* mask_b, int64_t N, int l2,
                                   const float* count_dev, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(g_total && mask_a && mask_b, "stage2_loss_bwd: null pointer");
    LossArgs a = {};
    a.g_total = g_total; a.count_dev = count_dev; a.rgb = rgb; a.rgb_gt = rgb_gt; a.L = L; a.k_rgb = k_rgb; a.d_rgb = d_rgb;
    a.alb = alb; a.alb_j = alb_j; a.k_alb = k_alb; a.d_alb = d_alb; a.d_alb_j = d_alb_j;
    a.wgt = wgt; a.wgt_j = wgt_j; a.nb = nb; a.k_wgt = k_wgt; a.d_wgt = d_wgt; a.d_wgt_j = d_wgt_j;
    a.vis = vis; a.vis_gt = vis_gt; a.V = V; a.k_vis = k_vis; a.d_vis = d_vis;
    a.nrm = nrm; a.nrm_gt = nrm_gt; a.nrm_j = nrm_j; a.k_nrm = k_nrm; a.k_nrmj = k_nrmj; a.d_nrm = d_nrm; a.d_nrm_j = d_nrm_j;
    a.mask_a = mask_a; a.mask_b = mask_b; a.N = N; a.l2 = l2;
    PSN_CHECK_ARG((d_rgb == nullptr || (rgb && rgb_gt)) && (d_alb == nullptr || (alb && alb_j && d_alb_j)) &&
                  (d_wgt == nullptr || (wgt && wgt_j && d_wgt_j)) && (d_vis == nullptr || (vis && vis_gt)) &&
                  (d_nrm == nullptr || (nrm && nrm_gt)), "stage2_loss_bwd: incomplete term");
    if (N <= 0) return PSN_OK;
    int64_t blocks = (N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const int by = (d_rgb != nullptr && L >= 2 * kLossChunksY) ? kLossChunksY : 1;
    hipLaunchKernelGGL(stage2_loss_bwd_kernel, dim3((unsigned)blocks, by), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("stage2_loss_bwd");
    return PSN_OK;
}

// ---- SparseAdam on the rows of the per-light tables (stage2/trainer.py:126-168, torch.optim.SparseAdam) ------------
// One thread per table row: the row moves iff it is among idx[0..n_idx) (the lights of this step; duplicates allowed),
// with torch's sparse_adam arithmetic: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); p += -step_size m / (sqrt(v) + eps).
namespace psn {
struct RowAdamArgs { PsnRowAdamItem it[PSN_ROW_ADAM_MAX]; const int64_t* idx; int n_idx; int n;
                     const float* dev; };  // dev: nullptr, or [n] step sizes ON THE DEVICE that replace it[i].step_size
__global__ __launch_bounds__(256) void row_adam_kernel(RowAdamArgs a) {
    const int item = blockIdx.y;
    if (item >= a.n) return;
    PsnRowAdamItem it = a.it[item];
    if (a.dev != nullptr) it.step_size = a.dev[item];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // (the index list through LDS in chunks of 256: a dependent global load per iteration otherwise)
    __shared__ int64_t sidx[256];
    bool touched = false;
    for (int i0 = 0; i0 < a.n_idx; i0 += 256) {
        const int m = a.n_idx - i0 < 256 ? a.n_idx - i0 : 256;
        __syncthreads();
        if ((int)threadIdx.x < m) sidx[threadIdx.x] = a.idx[i0 + threadIdx.x];
        __syncthreads();
        for (int j = 0; j < m; ++j) touched = touched || sidx[j] == r;
    }
    if (r >= it.rows || !touched) return;
    for (int c = 0; c < it.cols; ++c) {
        const int64_t e = r * it.cols + c;
        const float g = it.grad[e];
        float m = it.exp_avg[e], v = it.exp_avg_sq[e];
        m += (g - m) * it.one_minus_beta1;
        v += (g * g - v) * it.one_minus_beta2;
        it.exp_avg[e] = m;
        it.exp_avg_sq[e] = v;
        it.param[e] += (m / (sqrtf(v) + it.eps)) * (-it.step_size);
    }
}
}  // namespace psn

static int row_adam_impl(int n_items, const PsnRowAdamItem* items, const int64_t* idx, int n_idx, const float* step_sizes_dev, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(items && idx && n_items >= 1 && n_items <= PSN_ROW_ADAM_MAX && n_idx >= 0, "row_adam: bad arguments");
    RowAdamArgs a = {};
    a.dev = step_sizes_dev;
    int64_t max_rows = 0;
    for (int i = 0; i < n_items; ++i) {
        PSN_CHECK_ARG(items[i].param && items[i].grad && items[i].exp_avg && items[i].exp_avg_sq && items[i].rows >= 0 && items[i].cols >= 1,
                      "row_adam: item %d", i);
        a.it[i] = items[i];
        if (items[i].rows > max_rows) max_rows = items[i].rows;
    }
    a.idx = idx; a.n_idx = n_idx; a.n = n_items;
    if (max_rows <= 0 || n_idx == 0) return PSN_OK;
    hipLaunchKernelGGL(row_adam_kernel, dim3((unsigned)((max_rows + 255) / 256), n_items), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("row_adam");
    return PSN_OK;
}

extern "C" int psn_row_adam(int n_items, const PsnRowAdamItem* items, const int64_t* idx, int n_idx, void* stream) {
    return row_adam_impl(n_items, items, idx, n_idx, nullptr, stream);
}

// step sizes read from DEVICE memory (step_sizes_dev [n_items]): a launch that is identical from step to step (HIP-graph replay)
extern "C" int psn_row_adam_dev(int n_items, const PsnRowAdamItem* items, const int64_t* idx, int n_idx, const float* step_sizes_dev,
                                void* stream) {
    PSN_CHECK_ARG(step_sizes_dev != nullptr, "row_adam_dev: step_sizes_dev is required");
    return row_adam_impl(n_items, items, idx, n_idx, step_sizes_dev, stream);
}

constexpr int kPairSlices = 32;  // chunk slices of the final reduction when there are more than 128 chunks

extern "C" int64_t psn_pair_sums_group_workspace(int n_items, int V, int64_t Ns, int C) {
    int rows = 16;
    int64_t chunks = (Ns + rows - 1) / rows;
    while (chunks > PSN_PAIR_SUMS_MAX_CHUNKS) { rows *= 2; chunks = (Ns + rows - 1) / rows; }
    return (int64_t)n_items * ((chunks > 0 ? chunks : 1) + kPairSlices) * V * C;
}

extern "C" int psn_pair_sums_group(int n_items, const PsnPairSumsItem* items, int V, int64_t Ns, int C, const float* pe_l, int64_t ld_pe,
                                   int n_pe, int64_t ld_w, float* workspace, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(items && pe_l && workspace && n_items >= 1 && n_items <= PSN_PAIR_GROUP_MAX, "pair_sums_group: bad arguments (n_items=%d)", n_items);
    PSN_CHECK_ARG(V >= 1 && V <= PSN_PAIR_SUMS_MAX_V && C >= 4 && C <= 256 && C % 4 == 0 && Ns >= 0, "pair_sums_group: V=%d C=%d", V, C);
    PSN_CHECK_ARG(n_pe >= 1 && n_pe <= 64 && ld_pe >= n_pe && ld_w >= n_pe, "pair_sums_group: n_pe=%d", n_pe);
    PairGroupArgs a = {};
    for (int i = 0; i < n_items; ++i) {
        PSN_CHECK_ARG(items[i].x && items[i].sx && items[i].dWl && ((((uintptr_t)items[i].x | (uintptr_t)items[i].sx)) & 15) == 0, "pair_sums_group: item %d", i);
        a.it[i] = items[i];
    }
    int rows = 16;
    int64_t chunks = (Ns + rows - 1) / rows;
    while (chunks > PSN_PAIR_SUMS_MAX_CHUNKS) { rows *= 2; chunks = (Ns + rows - 1) / rows; }
    a.V = V; a.Ns = Ns; a.C = C; a.rows_per_block = rows; a.chunks = (int)chunks; a.part = workspace;
    a.pe_l = pe_l; a.ld_pe = ld_pe; a.n_pe = n_pe; a.ld_w = ld_w;
    PSN_CHECK_ARG((((uintptr_t)workspace) & 15) == 0, "pair_sums_group: workspace must be 16-byte aligned");
    if (chunks > 0) {
        hipLaunchKernelGGL(pair_group_sums_kernel, dim3((unsigned)chunks, n_items), dim3(256), 0, (hipStream_t)stream, a);
        PSN_CHECK_LAUNCH("pair_sums_group (sums)");
    }
    if (chunks > 128) {
        // sliced: 32 blocks per (item, column group) add a slice of the chunks each (fixed order), the last launch adds the slices
        a.slices = kPairSlices;
        a.part2 = workspace + (int64_t)n_items * chunks * V * C;
        hipLaunchKernelGGL(pair_group_final_kernel, dim3((unsigned)((C + 63) / 64), n_items, kPairSlices), dim3(1024), 0, (hipStream_t)stream, a);
        PSN_CHECK_LAUNCH("pair_sums_group (slices)");
        a.part = a.part2; a.chunks = kPairSlices; a.slices = 1; a.part2 = nullptr;
    }
    hipLaunchKernelGGL(pair_group_final_kernel, dim3((unsigned)((C + 63) / 64), n_items), dim3(1024), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("pair_sums_group (final)");
    return PSN_OK;
}

extern "C" int psn_pair_sums(const float* x, int V, int64_t Ns, int C, float* sx, float* sl_part, int* n_chunks, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(x && sx && sl_part && n_chunks, "pair_sums: null pointer");
    PSN_CHECK_ARG(V >= 1 && V <= PSN_PAIR_SUMS_MAX_V && C >= 1 && C <= 256 && Ns >= 0, "pair_sums: V=%d C=%d (V <= %d, C <= 256)", V, C, PSN_PAIR_SUMS_MAX_V);
    PSN_CHECK_ARG(C % 4 == 0 && (((uintptr_t)x | (uintptr_t)sx | (uintptr_t)sl_part) & 15) == 0, "pair_sums: C must be a multiple of 4, buffers 16-byte aligned");
    int rows = 32;
    int64_t chunks = (Ns + rows - 1) / rows;
    while (chunks * 4 > PSN_PAIR_SUMS_MAX_CHUNKS) { rows *= 2; chunks = (Ns + rows - 1) / rows; }
    *n_chunks = (int)chunks * 4;  // one partial per (block, row lane)
    if (Ns <= 0) return PSN_OK;
    hipLaunchKernelGGL(pair_sums_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, V, Ns, C, rows, sx, sl_part);
    PSN_CHECK_LAUNCH("pair_sums");
    return PSN_OK;
}
