// fp32 GEMM on v_mfma_f32_32x32x2_f32 with fused epilogues (training path of every MLP on the hot path).
// Reference ops replaced: torch.nn.Linear / F.relu / nn.Softplus(beta=100) and their autograd backward
// in stage1/model/network.py:85-106 and stage2/model/renderer.py:17-49.
//
// Workgroup tile 128 x BN x 16 with BN = 128 or 256 (256 only when the grid still covers the chip several times
// over, see psn_gemm), 4 waves in a 2x2 grid, each wave 64 x BN/2 = 2 x NT MFMA tiles.  fp32 MFMA has so little arithmetic per operand byte that a 128x128
// tile needs ~8.7 B/clk/CU of global loads at MFMA peak -- the measured per-CU streaming limit (~10 B/clk) --
// and PMC showed 35-53 % MFMA-busy for it; the 256-wide tile reads each activation row-panel once and halves
// the bytes per flop.  Operand tiles are staged k-major in LDS ([16][rows+4] floats) so that every MFMA
// operand read is a conflict-free ds_read_b32 of 32 consecutive floats; the next k-tile is fetched into
// registers while the current one is multiplied (one barrier per k-tile).  Numerics: exact fp32 fma chain
// per output element (k-ordered), see MI355X guide "FP32-input MFMA".
//
// Weight gradients (dW = dZ^T X, K = rows of the pass) use deterministic split-K; psn_gemm_tn_grouped runs all of
// them for one backward pass in a single launch and returns the bias gradients (column sums of dZ) as a by-product.
#include "common.h"

namespace psn {

constexpr int BM = 128, BK = 16, LDA = BM + 4, EPI_LD = 68;

struct GemmArgs {
    int64_t M;
    int N, K;
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    const float* bias;
    int epi;
    const float* aux_in;
    int64_t ld_aux_in;
    const float* aux_in2;
    int64_t ld_aux_in2;
    float* aux_out;
    int64_t ld_aux_out;
    int k_chunk;  // K range per blockIdx.z (multiple of BK)
    int tiles_n;
    int64_t n_tiles;  // tiles_m * tiles_n
    int split_k;
    int a_vec, b_vec;  // 16-byte vector loads allowed
    int c_vec, auxin_vec, auxin2_vec, auxout_vec;  // 16-byte vector epilogue accesses allowed
    int64_t split_stride;  // floats between split-K partial outputs (0 when split_k == 1)
    float* colsum;          // trans_a only, or nullptr: [split_k][M] sums over k of A[k, m] (bias gradient by-product)
    // optional second product accumulated into the same C (grouped weight gradients: C = A^T B + A2^T B2): the splits
    // seg_splits .. split_k-1 run over (A2, B2); seg_splits == split_k when there is no second segment
    const float* A2;
    int64_t lda2;
    const float* B2;
    int64_t ldb2;
    int seg_splits;
    int b_div, b_mod;  // b_div > 0 (trans_a, !trans_b only): row k of B is table row (k / b_div) % b_mod
    const float* Bt2;  // optional second table for the columns >= b_split (own row stride and index map)
    int64_t ldbt2;
    int b2_div, b2_mod, b_split;
};

// floor(k / d) for 0 <= k < 2^24, d >= 1: float reciprocal estimate, corrected by at most one.
__device__ __forceinline__ int fastdiv24(int k, int d) {
    int q = (int)((float)k * __builtin_amdgcn_rcpf((float)d));
    const int r = k - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
}

// Stage one 128 x 16 operand tile into registers.  KCONTIG: source rows run along k (row-major [rows][K]);
// otherwise the source is [K][rows] row-major.
template <bool KCONTIG, int ROWS>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t n_rows,
                                           int k0, int k_end, int vec_ok, int tid, float4 (&v)[ROWS / 64],
                                           int k_div = 0, int k_mod = 0, const float* __restrict__ src2 = nullptr,
                                           int64_t ld2 = 0, int k_div2 = 0, int k_mod2 = 0, int col_split = 0) {
#pragma unroll
    for (int u = 0; u < ROWS / 64; ++u) {
        int f = tid + 256 * u;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONTIG) {
            int r = f >> 2, kq = (f & 3) * 4;
            int64_t row = row0 + r;
            int k = k0 + kq;
            if (row < n_rows) {
                const float* p = src + row * ld + k;
                if (vec_ok && k + 3 < k_end) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < k_end) x.x = p[0];
                    if (k + 1 < k_end) x.y = p[1];
                    if (k + 2 < k_end) x.z = p[2];
                    if (k + 3 < k_end) x.w = p[3];
                }
            }
        } else {
            int kk = f / (ROWS / 4), rq = (f % (ROWS / 4)) * 4;
            int k = k0 + kk;
            int64_t row = row0 + rq;
            if (k < k_end) {
                // k_div > 0: the operand is a TABLE indexed like the fused kernel's (k / k_div) % k_mod (weight
                // gradients of an input block whose rows repeat per light or per point)
                // col_split > 0: the columns from col_split on come from a SECOND table with its own index map (the two
                // halves [PE(x_n) | PE(l_v)] of one input block, side by side in one product)
                const bool second = col_split > 0 && row >= col_split;
                const int kd = second ? k_div2 : k_div, km = second ? k_mod2 : k_mod;
                int ks = k;
                if (kd > 0) {
                    // (k / kd) % km without integer division (two of them per load were ~50 vector instructions per
                    // k-tile, and every one costs matrix-pipe time): float reciprocal + one correction step, exact for
                    // k < 2^24; km == 0 = the host found the modulo to be the identity ((K - 1) / kd < km)
                    if (k_end <= (1 << 24)) {
                        if (kd > 1) ks = fastdiv24(k, kd);
                        if (km > 0) ks -= fastdiv24(ks, km) * km;
                    } else {
                        ks = k / kd;
                        if (km > 0) ks %= km;
                    }
                }
                const float* p = second ? src2 + (int64_t)ks * ld2 + (row - col_split) : src + (int64_t)ks * ld + row;
                if (vec_ok && row + 3 < n_rows) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (row + 0 < n_rows) x.x = p[0];
                    if (row + 1 < n_rows) x.y = p[1];
                    if (row + 2 < n_rows) x.z = p[2];
                    if (row + 3 < n_rows) x.w = p[3];
                }
            }
        }
        v[u] = x;
    }
}

template <bool KCONTIG, int ROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, int tid, const float4 (&v)[ROWS / 64]) {
    constexpr int LD = ROWS + 4;
#pragma unroll
    for (int u = 0; u < ROWS / 64; ++u) {
        int f = tid + 256 * u;
        if (KCONTIG) {
            int r = f >> 2, kq = (f & 3) * 4;
            lds[(kq + 0) * LD + r] = v[u].x;
            lds[(kq + 1) * LD + r] = v[u].y;
            lds[(kq + 2) * LD + r] = v[u].z;
            lds[(kq + 3) * LD + r] = v[u].w;
        } else {
            int kk = f / (ROWS / 4), rq = (f % (ROWS / 4)) * 4;
            *reinterpret_cast<float4*>(&lds[kk * LD + rq]) = v[u];
        }
    }
}

template <int NT>
struct GemmLds {
    static constexpr int BN = 64 * NT, LDB = BN + 4;
    static constexpr int A_FLOATS = BK * LDA, B_FLOATS = BK * LDB, BUF_FLOATS = A_FLOATS + B_FLOATS;
    // operand tiles [buf]{A[BK][LDA], B[BK][LDB]}; re-used by the epilogue as 4 wave-private [32][EPI_LD] tiles
    static constexpr int FLOATS = 2 * BUF_FLOATS > 4 * 32 * EPI_LD ? 2 * BUF_FLOATS : 4 * 32 * EPI_LD;
};

// One workgroup = one (output tile, split-K slice) work item `bid` of the problem g.
template <bool TA, bool TB, int NT>  // NT = 32-column MFMA tiles per wave in N: BN = 64 * NT
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int64_t bid, float* lds_raw) {
    constexpr int BN = GemmLds<NT>::BN, LDB = GemmLds<NT>::LDB;
    constexpr int A_FLOATS = GemmLds<NT>::A_FLOATS, BUF_FLOATS = GemmLds<NT>::BUF_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware work order.  The dispatcher places block b on XCD b % 8 (per-XCD L2s).  Work items are
    // numbered split-major / tile-minor and each XCD gets a contiguous run of them, so the tiles that share
    // an operand panel -- the n-tiles of one A row-panel, or all output tiles of one split-K slice (which
    // stream the same K-chunk of both operands) -- run on the same L2 at about the same time.  Bijective
    // remap (guide T1) over the 1-D grid of n_tiles * split_k blocks.
    const int64_t nb = g.n_tiles * (int64_t)g.split_k;
    int64_t q = nb / 8, r8 = nb % 8, xcd = bid % 8, idx = bid / 8;
    int64_t w = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int split = (int)(w / g.n_tiles);
    const int64_t t = w % g.n_tiles;
    const int64_t tm = t / g.tiles_n;
    const int tn = (int)(t % g.tiles_n);
    const int64_t m0 = tm * BM;
    const int n0 = tn * BN;
    const bool seg2 = split >= g.seg_splits;  // second (A2, B2) product of a grouped weight gradient
    const float* Ap = seg2 ? g.A2 : g.A;
    const float* Bp = seg2 ? g.B2 : g.B;
    const int64_t lda = seg2 ? g.lda2 : g.lda, ldb = seg2 ? g.ldb2 : g.ldb;
    const int k_begin = (seg2 ? split - g.seg_splits : split) * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);

    floatx16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register-staged double buffering: while tile t is multiplied out of LDS buffer t&1, tile t+1 is fetched
    // into registers and written to the other buffer after the MFMA stream (one barrier per k-tile).
    float4 ra0[2], rb0[BN / 64];
    const int nt = (k_end - k_begin + BK - 1) / BK;
    // Bias-gradient by-product of dW = dZ^T X: with A = dZ stored [K][M], thread tid always stages the same four
    // columns m0 + 4 (tid % 32) .. +3 (rows k = tid / 32 and tid / 32 + 8 of every k-tile), so their sum over k is one
    // float4 per thread; only the n-tile-0 workgroups of each row panel keep it.
    const bool do_cs = TA && g.colsum != nullptr && tn == 0 && !seg2;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define PSN_FETCH(T)                                                                                    \
    fetch_tile<!TA, BM>(Ap, lda, m0, g.M, k_begin + (T) * BK, k_end, g.a_vec, tid, ra0);               \
    fetch_tile<TB, BN>(Bp, ldb, n0, g.N, k_begin + (T) * BK, k_end, g.b_vec, tid, rb0, g.b_div, g.b_mod, g.Bt2, g.ldbt2,   \
                       g.b2_div, g.b2_mod, g.b_split);
#define PSN_STORE(BUF)                                                   \
    if (TA && do_cs) {                                                   \
        cs.x += ra0[0].x + ra0[1].x; cs.y += ra0[0].y + ra0[1].y;        \
        cs.z += ra0[0].z + ra0[1].z; cs.w += ra0[0].w + ra0[1].w;        \
    }                                                                    \
    store_tile<!TA, BM>(lds_raw + (BUF) * BUF_FLOATS, tid, ra0);         \
    store_tile<TB, BN>(lds_raw + (BUF) * BUF_FLOATS + A_FLOATS, tid, rb0);
// MFMA operands of k-step j+1 are read from LDS while the MFMAs of step j execute (two register sets, order
// given to the scheduler with sched_group_barrier); hipcc otherwise re-uses one operand register set per k-step
// and serialises {ds_read, s_waitcnt lgkmcnt(0), MFMAs}.
#define PSN_COMPUTE(BUF)                                                                         \
    {                                                                                            \
        const float* As = lds_raw + (BUF) * BUF_FLOATS;                                          \
        const float* Bs = As + A_FLOATS;                                                         \
        float pa[8][2], pb[8][NT];                                                               \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) pa[j][i] = As[(2 * j + lh) * LDA + wr * 64 + i * 32 + li];             \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) pb[j][n] = Bs[(2 * j + lh) * LDB + wc * (32 * NT) + n * 32 + li];     \
        }                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                        \
                _Pragma("unroll") for (int n = 0; n < NT; ++n)                                   \
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j][i], pb[j][n], acc[i][n], 0, 0, 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x100, (2 + NT) / 2, 0);                            \
        _Pragma("unroll") for (int j = 0; j < 7; ++j) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, (2 + NT) / 2, 0);                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);                              \
        }                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);                                  \
    }
    PSN_FETCH(0)
    PSN_STORE(0)
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) { PSN_FETCH(t + 1) }
        __builtin_amdgcn_sched_barrier(0);  // global loads are issued BEFORE the pinned MFMA / LDS-read stream
        PSN_COMPUTE(buf)
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) { PSN_STORE(buf ^ 1) }
        __syncthreads();
    }
#undef PSN_FETCH
#undef PSN_STORE
#undef PSN_COMPUTE

    if (TA && do_cs) {  // fixed-order reduction over the 8 threads that share a column group: deterministic
        cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64);
        cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
        float4* red = reinterpret_cast<float4*>(lds_raw);  // the operand tiles are dead (barrier at the loop end)
        if (lane < 32) red[wave * 32 + lane] = cs;
        __syncthreads();
        if (tid < 32) {
            float4 a = red[tid], b = red[32 + tid], c = red[64 + tid], d = red[96 + tid];
            float* dst = g.colsum + (int64_t)split * g.M + m0 + 4 * tid;
            const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (m0 + 4 * tid + e < g.M) dst[e] = v[e];
        }
        __syncthreads();  // the epilogue re-uses the same LDS
    }

    // epilogue.  Lane (j = li, h = lh) holds C[m0 + wr*64 + it*32 + (r&3) + 8*(r>>2) + 4*h][n0 + wc*64 + jt*32 + j].
    // Each wave transposes one 32 x 64 half of its tile through a private LDS tile and then walks it in
    // float4 row segments (16 lanes = one 256 B row piece): coalesced 16-byte loads/stores for C and the
    // aux operands, and a small rolled loop so the fused epilogue math does not inflate register use.
    float* Cbase = g.C + (int64_t)split * g.split_stride;
    float* et = lds_raw + wave * (32 * EPI_LD);
    const int c4 = (lane & 15) * 4;          // column (within a 64-column group) of this lane's float4
#pragma unroll
    for (int cg = 0; cg < NT / 2; ++cg) {    // 64-column groups of the wave's tile
    const int nbase = n0 + wc * (32 * NT) + cg * 64 + c4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias != nullptr) {
        if (nbase + 0 < g.N) bias4.x = g.bias[nbase + 0];
        if (nbase + 1 < g.N) bias4.y = g.bias[nbase + 1];
        if (nbase + 2 < g.N) bias4.z = g.bias[nbase + 2];
        if (nbase + 3 < g.N) bias4.w = g.bias[nbase + 3];
    }
    const bool full4 = nbase + 3 < g.N;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                et[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_LD + jt * 32 + li] = acc[it][2 * cg + jt][r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int row = (lane >> 4) + 4 * i;
            const int64_t m = m0 + wr * 64 + it * 32 + row;
            if (m >= g.M || nbase >= g.N) continue;
            float4 v = *reinterpret_cast<const float4*>(&et[row * EPI_LD + c4]);
            float vv[4] = {v.x, v.y, v.z, v.w};
            const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
            float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f}, o2[4] = {0.f, 0.f, 0.f, 0.f};
            float cin[4] = {0.f, 0.f, 0.f, 0.f};
            const int epi = g.epi;
            const bool need_a1 = epi == PSN_EPI_MUL_AUX || epi == PSN_EPI_MUL_POS || epi >= PSN_EPI_MUL2;
            const bool need_a2 = epi == PSN_EPI_MUL2 || epi == PSN_EPI_SOFTPLUS_BWD;
            const bool has_o2 = (epi == PSN_EPI_BIAS_SOFTPLUS && g.aux_out != nullptr) || epi == PSN_EPI_MUL2 ||
                                epi == PSN_EPI_MUL_AUX_RAW;
            float* cp = Cbase + m * g.ldc + nbase;
            if (need_a1) {
                const float* ap = g.aux_in + m * g.ld_aux_in + nbase;
                if (full4 && g.auxin_vec) { float4 t = *reinterpret_cast<const float4*>(ap); a1[0] = t.x; a1[1] = t.y; a1[2] = t.z; a1[3] = t.w; }
                else { for (int e = 0; e < 4; ++e) if (nbase + e < g.N) a1[e] = ap[e]; }
            }
            if (need_a2) {
                const float* ap = g.aux_in2 + m * g.ld_aux_in2 + nbase;
                if (full4 && g.auxin2_vec) { float4 t = *reinterpret_cast<const float4*>(ap); a2[0] = t.x; a2[1] = t.y; a2[2] = t.z; a2[3] = t.w; }
                else { for (int e = 0; e < 4; ++e) if (nbase + e < g.N) a2[e] = ap[e]; }
            }
            if (epi == PSN_EPI_ACCUM) {
                if (full4 && g.c_vec) { float4 t = *reinterpret_cast<const float4*>(cp); cin[0] = t.x; cin[1] = t.y; cin[2] = t.z; cin[3] = t.w; }
                else { for (int e = 0; e < 4; ++e) if (nbase + e < g.N) cin[e] = cp[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = vv[e];
                switch (epi) {
                    case PSN_EPI_NONE: break;
                    case PSN_EPI_BIAS: x = x + bb[e]; break;
                    case PSN_EPI_BIAS_RELU: x = fmaxf(x + bb[e], 0.0f); break;
                    case PSN_EPI_BIAS_SOFTPLUS: { float sp, sg; softplus100_sig(x + bb[e], sp, sg); x = sp; o2[e] = sg; break; }
                    case PSN_EPI_MUL_AUX: x = x * a1[e]; break;
                    case PSN_EPI_MUL_POS: x = a1[e] > 0.0f ? x : 0.0f; break;
                    case PSN_EPI_BIAS_SIGMOID: x = sigmoidf_(x + bb[e]); break;
                    case PSN_EPI_ACCUM: x = cin[e] + x; break;
                    case PSN_EPI_MUL2: o2[e] = x * a2[e]; x = x * a1[e]; break;
                    case PSN_EPI_SOFTPLUS_BWD: x = a1[e] * (x + 100.0f * a2[e] * (1.0f - a1[e])); break;
                    case PSN_EPI_MUL_AUX_RAW: o2[e] = x; x = x * a1[e]; break;
                    default: break;
                }
                vv[e] = x;
            }
            if (full4 && g.c_vec) *reinterpret_cast<float4*>(cp) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            else { for (int e = 0; e < 4; ++e) if (nbase + e < g.N) cp[e] = vv[e]; }
            if (has_o2) {
                float* op = g.aux_out + m * g.ld_aux_out + nbase;
                if (full4 && g.auxout_vec) *reinterpret_cast<float4*>(op) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                else { for (int e = 0; e < 4; ++e) if (nbase + e < g.N) op[e] = o2[e]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }  // cg
}

template <bool TA, bool TB, int NT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds_raw[GemmLds<NT>::FLOATS];
    gemm_tile<TA, TB, NT>(g, blockIdx.x, lds_raw);
}

// Grouped weight gradients: up to kMaxGroup independent C_i = A_i^T B_i (+ A2_i^T B2_i) problems that share K (the row
// count of one backward pass) in ONE launch.  A single 256 x 256 gradient has only four output tiles, so on its own it
// needs ~128 split-K slices to fill 256 CUs and then spends as long reducing 128 partial tiles as multiplying; sixteen
// of them together fill the chip with 8 slices each.  Block ranges are padded to multiples of 8 so that the XCD-aware
// work order of gemm_tile stays aligned with the hardware's block -> XCD round robin.
constexpr int kMaxGroup = 12;
struct GroupedArgs {
    int n;
    int64_t block_start[kMaxGroup + 1];
    GemmArgs g[kMaxGroup];
};
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(GroupedArgs gg) {
    __shared__ __attribute__((aligned(16))) float lds_raw[GemmLds<2>::FLOATS];
    int i = 0;
    while (i + 1 < gg.n && (int64_t)blockIdx.x >= gg.block_start[i + 1]) ++i;
    const GemmArgs g = gg.g[i];
    const int64_t bid = (int64_t)blockIdx.x - gg.block_start[i];
    if (bid >= g.n_tiles * (int64_t)g.split_k) return;  // padding block
    gemm_tile<true, false, 2>(g, bid, lds_raw);
}

// Branch-free staging of one 16 x 256 operand tile of the one-tile-per-workgroup kernel (source [K][rows], row-major,
// 16-byte aligned, ld % 4 == 0): addresses are clamped into the allocation and out-of-range elements zeroed with
// selects, so all loads of a k-tile sit in one basic block and the waits on them can be counted (vmcnt(N)) instead of
// draining the queue -- the condition for keeping two tiles in flight.
template <int NU>
__device__ __forceinline__ void fetch_tile256(const float* __restrict__ src, int64_t ld, int n_rows, int k0, int k_end,
                                              int tid, float4 (&v)[NU]) {
    // clamp to the last 4-column group of the logical matrix: with a 16-byte aligned base and ld % 4 == 0 that group
    // lies inside the row of the underlying buffer even when the operand is a column slice of it
    const int rqc = min((tid & 63) * 4, ((n_rows - 1) >> 2) << 2);
    // the k row of a load is wave-uniform (tid >> 6 = wave): scalar row pointer + one constant lane offset, so the eight
    // loads of a k-tile cost no vector address arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int kc = min(k0 + wave + 4 * u, k_end - 1);
        const float* row = src + (int64_t)kc * ld;
        v[u] = *reinterpret_cast<const float4*>(row + rqc);
    }
}
// ... the zeroing happens when the tile is written to LDS (two k-tiles later), never right behind the loads
template <int NU>
__device__ __forceinline__ void mask_tile256(int n_rows, int k0, int k_end, int tid, float4 (&v)[NU]) {
    const int rq = (tid & 63) * 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool c0 = rq + 0 < n_rows, c1 = rq + 1 < n_rows, c2 = rq + 2 < n_rows, c3 = rq + 3 < n_rows;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bool kin = k0 + wave + 4 * u < k_end;  // wave-uniform
        v[u].x = (kin && c0) ? v[u].x : 0.f;
        v[u].y = (kin && c1) ? v[u].y : 0.f;
        v[u].z = (kin && c2) ? v[u].z : 0.f;
        v[u].w = (kin && c3) ? v[u].w : 0.f;
    }
}

// ---- weight gradients with ONE 256 x 256 tile per workgroup ----------------------------------------------------------
// dW = dZ^T X with M, N <= 256: a workgroup owns the whole output for one K slice, so every operand row is read from
// L2 exactly once (the 128 x 128 tiles above read each twice and sit at the per-CU streaming limit).  4 waves in a
// 2 x 2 grid, each 128 x 128 = 4 x 4 MFMA tiles (256 accumulator registers, one wave per SIMD); per 32-row k-tile a
// wave issues 256 MFMAs (16384 cycles) against 128 ds_read_b32, 16 staged float4 loads and ONE barrier (16-row tiles:
// a barrier and an exposed first fragment read per 8192 cycles).
constexpr int T256 = 256, T256_LD = T256 + 4, TK = 32, T256_BUF = 2 * TK * T256_LD;  // k-tiles of TK = 32 rows: 2 x 66.5 KB of LDS
// The k-tiles a step stages need no masks except the last two of a K slice (the possibly partial final tile and the
// overshoot of the prefetch), so the steady-state loop runs mask-free steps and only the last <= 3 steps of a slice
// the masked ones.  The 32 v_cndmask per k-tile of the general variant cost 4.6 % of the kernel (timing-only build:
// 129 -> 135 TF incl. the reduction on 16 products at K = 524,288; every vector instruction is paid for in fp32-MFMA time).
__global__ __launch_bounds__(256, 1) void gemm_tn256_grouped_kernel(GroupedArgs gg) {
    extern __shared__ __attribute__((aligned(16))) float lds256[];  // 2 x {A[32][260], B[32][260]}
    int gi = 0;
    while (gi + 1 < gg.n && (int64_t)blockIdx.x >= gg.block_start[gi + 1]) ++gi;
    const GemmArgs g = gg.g[gi];
    const int split = (int)((int64_t)blockIdx.x - gg.block_start[gi]);
    if (split >= g.split_k) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const bool seg2 = split >= g.seg_splits;
    const float* Ap = seg2 ? g.A2 : g.A;
    const float* Bp = seg2 ? g.B2 : g.B;
    const int64_t lda = seg2 ? g.lda2 : g.lda, ldb = seg2 ? g.ldb2 : g.ldb;
    const int k_begin = (seg2 ? split - g.seg_splits : split) * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nt = (k_end - k_begin + TK - 1) / TK;

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
    // Two register sets: the operand rows of k-tiles t+2 and t+3 are in flight while tile t is multiplied.  With one
    // workgroup per CU and nothing re-used between workgroups, HBM has to deliver ~2.3 TB/s for the matrix pipe to stay
    // busy; a single 32 KB tile in flight per CU (8 MB on the chip) only sustains about half of that.
    float4 ra[2][8], rb[2][8];
    const bool do_cs = g.colsum != nullptr && !seg2;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define T_FETCH(T, S)                                                                            \
    fetch_tile256(Ap, lda, (int)g.M, k_begin + (T) * TK, k_end, tid, ra[S]);                     \
    fetch_tile256(Bp, ldb, g.N, k_begin + (T) * TK, k_end, tid, rb[S]);
#define T_STORE(BUF, S, T)                                                                       \
    mask_tile256((int)g.M, k_begin + (T) * TK, k_end, tid, ra[S]);                               \
    mask_tile256(g.N, k_begin + (T) * TK, k_end, tid, rb[S]);                                    \
    if (do_cs) {                                                                                 \
        _Pragma("unroll") for (int u2_ = 0; u2_ < 8; u2_ += 2) {                                 \
            cs.x += ra[S][u2_].x + ra[S][u2_ + 1].x; cs.y += ra[S][u2_].y + ra[S][u2_ + 1].y;    \
            cs.z += ra[S][u2_].z + ra[S][u2_ + 1].z; cs.w += ra[S][u2_].w + ra[S][u2_ + 1].w;    \
        }                                                                                        \
    }                                                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) {                                           \
        *reinterpret_cast<float4*>(lds256 + (BUF) * T256_BUF + ((tid >> 6) + 4 * u_) * T256_LD + (tid & 63) * 4) = ra[S][u_]; \
        *reinterpret_cast<float4*>(lds256 + (BUF) * T256_BUF + TK * T256_LD + ((tid >> 6) + 4 * u_) * T256_LD + (tid & 63) * 4) = rb[S][u_]; \
    }
    // Staging item J (0..15) of a step: one float4 of operand A (J < 8, k row wave + 4 J) or B (k row wave + 4 (J - 8)) of
    // tile T+1 is masked, added to the column sums (A; same association as T_STORE), written to the other LDS buffer,
    // and its register receives the same row of tile T+3.  One item rides in the MFMA gaps of each k-pair of T_STEP.
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rq_ = (tid & 63) * 4;
    const bool am0 = rq_ + 0 < (int)g.M, am1 = rq_ + 1 < (int)g.M, am2 = rq_ + 2 < (int)g.M, am3 = rq_ + 3 < (int)g.M;
    const bool bm0 = rq_ + 0 < g.N, bm1 = rq_ + 1 < g.N, bm2 = rq_ + 2 < g.N, bm3 = rq_ + 3 < g.N;
    const int rqa = min(rq_, (((int)g.M - 1) >> 2) << 2), rqb = min(rq_, ((g.N - 1) >> 2) << 2);
    float4 cs01 = make_float4(0.f, 0.f, 0.f, 0.f);
#define T_ITEM(BUF, S, T, J, MASKED, BD)                                                         \
    if (!((BD) && (J) >= 8)) {                                                                   \
        constexpr int u_ = (J) & 7;                                                              \
        constexpr bool isb_ = (J) >= 8;                                                          \
        float4& x_ = isb_ ? rb[S][u_] : ra[S][u_];                                               \
        if (MASKED) {                                                                            \
            const bool kin_ = k_begin + ((T) + 1) * TK + wave_u + 4 * u_ < k_end;                \
            x_.x = (kin_ && (isb_ ? bm0 : am0)) ? x_.x : 0.f;                                    \
            x_.y = (kin_ && (isb_ ? bm1 : am1)) ? x_.y : 0.f;                                    \
            x_.z = (kin_ && (isb_ ? bm2 : am2)) ? x_.z : 0.f;                                    \
            x_.w = (kin_ && (isb_ ? bm3 : am3)) ? x_.w : 0.f;                                    \
        }                                                                                        \
        if (!isb_ && do_cs) {                                                                    \
            if ((u_ & 1) == 0) cs01 = x_;                                                        \
            else { cs.x += cs01.x + x_.x; cs.y += cs01.y + x_.y; cs.z += cs01.z + x_.z; cs.w += cs01.w + x_.w; } \
        }                                                                                        \
        *reinterpret_cast<float4*>(lds256 + (BUF) * T256_BUF + (isb_ ? TK * T256_LD : 0) + (wave_u + 4 * u_) * T256_LD + rq_) = x_; \
        const int kc_ = min(k_begin + ((T) + 3) * TK + wave_u + 4 * u_, k_end - 1);             \
        x_ = *reinterpret_cast<const float4*>((isb_ ? Bp + (int64_t)kc_ * ldb + rqb : Ap + (int64_t)kc_ * lda + rqa)); \
    }
    // one wave per SIMD: nothing else hides latencies or the staging work, so every k-pair j of a step is its own
    // scheduling region: operand reads of k-pair j+1 first, then the 16 MFMAs of k-pair j with staging item j in their gaps
#define T_STEP(T, S, MASKED, BD)                                                                 \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (BD) { T_DMA_B(((T) + 1) & 1, (T) + 1) __builtin_amdgcn_sched_barrier(0); }           \
        const float* As = lds256 + ((T) & 1) * T256_BUF + wr * 128 + li;                         \
        const float* Bs = lds256 + ((T) & 1) * T256_BUF + TK * T256_LD + wc * 128 + li;          \
        float pa[2][4], pb[2][4];                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) pa[0][i] = As[lh * T256_LD + i * 32];      \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) pb[0][n] = Bs[lh * T256_LD + n * 32];      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        T_PAIR(T, S, 0, MASKED, BD) T_PAIR(T, S, 1, MASKED, BD) T_PAIR(T, S, 2, MASKED, BD) T_PAIR(T, S, 3, MASKED, BD)  \
        T_PAIR(T, S, 4, MASKED, BD) T_PAIR(T, S, 5, MASKED, BD) T_PAIR(T, S, 6, MASKED, BD) T_PAIR(T, S, 7, MASKED, BD)  \
        T_PAIR(T, S, 8, MASKED, BD) T_PAIR(T, S, 9, MASKED, BD) T_PAIR(T, S, 10, MASKED, BD) T_PAIR(T, S, 11, MASKED, BD)  \
        T_PAIR(T, S, 12, MASKED, BD) T_PAIR(T, S, 13, MASKED, BD) T_PAIR(T, S, 14, MASKED, BD) T_PAIR(T, S, 15, MASKED, BD)  \
        /* this wave's LDS-DMA rows of B tile T + 1 (the 8 oldest outstanding requests) have landed; the 8 younger operand-A \
           rows of tile T + 3 stay in flight */                                                  \
        if (BD) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                 \
        lds_barrier(); /* NOT __syncthreads(): the operand rows of tile T + 3 stay in flight across the barrier */ \
    }
#define T_PAIR(T, S, J, MASKED, BD)                                                              \
    {                                                                                            \
        if ((J) < 15) {                                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) pa[((J) + 1) & 1][i] = As[(2 * (J) + 2 + lh) * T256_LD + i * 32]; \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) pb[((J) + 1) & 1][n] = Bs[(2 * (J) + 2 + lh) * T256_LD + n * 32]; \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int n = 0; n < 4; ++n)                                        \
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[(J) & 1][i], pb[(J) & 1][n], acc[i][n], 0, 0, 0); \
        T_ITEM(((T) + 1) & 1, S, T, J, MASKED, BD)                                               \
        if ((J) < 15) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                         \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                   \
            if (q_ == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                      \
            if (q_ == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
    // Operand B of a product that is exactly 256 x 256 arrives by LDS-DMA (global_load_lds, one 1 KB k row per instruction,
    // 8 per wave and k-tile, issued at the start of the step that multiplies the PREVIOUS tile): it has no column sums to feed
    // and needs no masks -- rows behind k_end are clamped duplicates that meet zeroed rows of A --, so its 64 staging
    // registers, 8 ds_write_b128 and 8 vector loads per step and their waits disappear from the MFMA stream.
#define T_DMA_B(BUF, T)                                                                          \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) {                                           \
        const int kc_ = min(k_begin + (T) * TK + wave_u + 4 * u_, k_end - 1);                    \
        lds_dma_16<0>(Bp + (int64_t)kc_ * ldb, lds_addr(lds256 + (BUF) * T256_BUF + TK * T256_LD + (wave_u + 4 * u_) * T256_LD), \
                      (unsigned)(tid & 63) * 16u);                                               \
    }
    // Column masks are not needed for correctness in either path: operand columns >= M / >= N are loaded from clamped
    // (in-bounds) addresses and only ever meet output rows / columns that are not stored.  What must be zeroed are the k rows
    // behind k_end (clamped duplicates), and only the last <= 3 steps of a slice stage such rows: the steady state of EVERY
    // product runs the mask-free step (a 217-wide product used to run the masked step throughout: 76 vs 128 TFLOP/s).
    int t = 0;
    if (g.N == T256) {  // (workgroup-uniform) full-width rows of B: LDS-DMA
        T_DMA_B(0, 0)
        fetch_tile256(Ap, lda, (int)g.M, k_begin, k_end, tid, ra[0]);
        mask_tile256((int)g.M, k_begin, k_end, tid, ra[0]);
        if (do_cs) {
#pragma unroll
            for (int u2_ = 0; u2_ < 8; u2_ += 2) {
                cs.x += ra[0][u2_].x + ra[0][u2_ + 1].x; cs.y += ra[0][u2_].y + ra[0][u2_ + 1].y;
                cs.z += ra[0][u2_].z + ra[0][u2_ + 1].z; cs.w += ra[0][u2_].w + ra[0][u2_ + 1].w;
            }
        }
#pragma unroll
        for (int u_ = 0; u_ < 8; ++u_)
            *reinterpret_cast<float4*>(lds256 + ((tid >> 6) + 4 * u_) * T256_LD + (tid & 63) * 4) = ra[0][u_];
        fetch_tile256(Ap, lda, (int)g.M, k_begin + TK, k_end, tid, ra[0]);
        fetch_tile256(Ap, lda, (int)g.M, k_begin + 2 * TK, k_end, tid, ra[1]);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // the DMA rows of tile 0 (older than the 16 operand-A rows just requested)
        __syncthreads();
        // steps <= nt - 3 stage tiles <= nt - 2: complete 16 x 256 tiles, no masks
        for (; t + 3 < nt; t += 2) {
            T_STEP(t, 0, false, true)
            T_STEP(t + 1, 1, false, true)
        }
        for (; t + 1 < nt; t += 2) {  // no control flow around the MFMA blocks: 256 accumulators must not meet a merge
            T_STEP(t, 0, true, true)
            T_STEP(t + 1, 1, true, true)
        }
        if (t < nt) T_STEP(t, 0, true, true)
    } else {
        T_FETCH(0, 0)
        T_STORE(0, 0, 0)
        T_FETCH(1, 0)
        T_FETCH(2, 1)
        __syncthreads();
        for (; t + 3 < nt; t += 2) {
            T_STEP(t, 0, false, false)
            T_STEP(t + 1, 1, false, false)
        }
        for (; t + 1 < nt; t += 2) {
            T_STEP(t, 0, true, false)
            T_STEP(t + 1, 1, true, false)
        }
        if (t < nt) T_STEP(t, 0, true, false)
    }
#undef T_DMA_B
#undef T_STEP
#undef T_PAIR
#undef T_ITEM
#undef T_FETCH
#undef T_STORE
    if (do_cs) {  // thread tid staged columns 4 (tid % 64) .. +3 (rows tid / 64 + 4u of every k-tile): reduce over the 4 waves
        float4* red = reinterpret_cast<float4*>(lds256);
        red[wave * 64 + lane] = cs;
        __syncthreads();
        if (tid < 64) {
            const float4 a = red[tid], b = red[64 + tid], c = red[128 + tid], d = red[192 + tid];
            const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
            float* dst = g.colsum + (int64_t)split * g.M + 4 * tid;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * tid + e < g.M) dst[e] = v[e];
        }
    }
    // partial tile -> workspace.  Lane (li, lh) holds C[wr*128 + i*32 + (r&3) + 8*(r>>2) + 4*lh][wc*128 + n*32 + li].
    float* Cw = g.C + (int64_t)split * g.split_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = wc * 128 + n * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) Cw[(int64_t)row * g.N + col] = acc[i][n][r];
            }
        }
}

// ---- the same product on the bf16 matrix pipe with split operands ("bf16x6"; EXPERIMENT, never the default) ------------
// dW = dZ^T X for 128 < M, N <= 256 with every fp32 operand element as the exact sum of three bf16 numbers (x = hi + mid + lo,
// 8 + 8 + 8 significant bits, round-to-nearest pieces) and a product as the six partial products of weight >= 2^-16,
//     lo hi + hi lo + mid mid + mid hi + hi mid + hi hi   (fp32 accumulation, smallest terms first),
// i.e. fp32-class results (relative error of a product ~2^-22) at 16 / 6 = 2.7x the fp32-MFMA rate.  At that rate the kernel is
// HBM-bound: per 16-row k-tile a workgroup reads 32 KB and issues 4 x 96 v_mfma_f32_32x32x16_bf16 (3072 matrix cycles per wave,
// 1.28 us), which asks ~6.4 TB/s of the chip -- so the job is to read every operand row exactly once (one 256 x 256 tile per
// workgroup, as above) and to keep two k-tiles of rows in flight per CU.
// Both operands are K-major in memory ([K][256] rows) while the MFMA wants 8 consecutive k per lane for its m / n: a thread stages
// the 4 x 4 block (k = 4 wave + j, m = lane + 64 e) of a tile, splits it and writes, per m and plane, the four k as one 8-byte
// LDS word into a [256 m][16 k] bf16 plane with 48-byte rows (fragment reads: one conflict-free ds_read_b128 per 32 x 16 operand
// tile and plane; the staging writes are 2-way conflicted).  2 x (2 operands x 3 planes x 12 KB) = 144 KB of LDS, one workgroup per CU.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
typedef float gfloatx2 __attribute__((ext_vector_type(2)));
#ifndef X3_SCHED
#define X3_SCHED 1
#endif
#ifndef X3_DO_MFMA   // (timing-only builds of tools/dbg: which part of a step the kernel waits for)
#define X3_DO_MFMA 1
#define X3_DO_STAGE 1
#define X3_DO_FETCH 1
#endif
constexpr int XK = 16, X_ROW = 48, X_PLANE = T256 * X_ROW, X_OPND = 3 * X_PLANE, X_BUF = 2 * X_OPND;  // bytes
__device__ __forceinline__ int x3g_cvt2(float a, float b) {
    gfloatx2 f;
    f[0] = a; f[1] = b;
    return __builtin_bit_cast(int, __builtin_convertvector(f, gbf16x2));  // v_cvt_pk_bf16_f32 (round to nearest even): a in the low half
}
__device__ __forceinline__ void x3g_split2(float a, float b, int& hi, int& mid, int& lo) {
    hi = x3g_cvt2(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & (int)0xFFFF0000);
    mid = x3g_cvt2(ra, rb);
    lo = x3g_cvt2(ra - __builtin_bit_cast(float, mid << 16), rb - __builtin_bit_cast(float, mid & (int)0xFFFF0000));
}
// Eight waves, TWO per SIMD (measured with four waves of 128 x 128 each, one per SIMD: the vector / LDS / memory work of a step
// and its MFMAs add up exactly -- 2.85 ms + 0.67 ms per partial product at K = 537k x 16 products -- a wave's own MFMAs do not
// hide its other instructions; a second wave on the SIMD does): wave (wr, wc) owns 128 x 64 of the tile (4 x 2 MFMA tiles, 128
// accumulator registers) and stages the 4 x 2 block (k = 4 (wave & 3) + j, m = lane + 64 (2 (wave >> 2) + e)) of every k-tile.
// NP = partial products per multiply: 6 (three pieces per operand: fp32-class), 3 (two pieces: hi hi + hi mid + mid hi, ~16 significant
// bits) or 1 (plain bf16 operands, fp32 accumulation).  Pieces that no product reads are neither formed nor written.
template <int NP>
__global__ __launch_bounds__(512, 1) void gemm_tn256_x3_grouped_kernel(GroupedArgs gg) {
    constexpr int NPL = NP == 6 ? 3 : NP == 3 ? 2 : 1;  // planes per operand
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsx[];  // 2 x {A planes hi/mid/lo, B planes hi/mid/lo}
    int gi = 0;
    while (gi + 1 < gg.n && (int64_t)blockIdx.x >= gg.block_start[gi + 1]) ++gi;
    const GemmArgs g = gg.g[gi];
    const int split = (int)((int64_t)blockIdx.x - gg.block_start[gi]);
    if (split >= g.split_k) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int kb = wave & 3, half = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    const bool seg2 = split >= g.seg_splits;
    const float* Ap = seg2 ? g.A2 : g.A;
    const float* Bp = seg2 ? g.B2 : g.B;
    const int64_t lda = seg2 ? g.lda2 : g.lda, ldb = seg2 ? g.ldb2 : g.ldb;
    const int k_begin = (seg2 ? split - g.seg_splits : split) * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nt = (k_end - k_begin + XK - 1) / XK;

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
    // staging registers: two k-tiles in flight (tile u lives in set (u + 1) & 1); [e][j]: column lane + 64 (2 half + e), k row 4 kb + j
    float ra[2][2][4], rb[2][2][4];
    int ca[2], cb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {  // columns behind M / N are read from clamped (in-bounds) addresses: they only meet outputs that are not stored
        ca[e] = min(lane + 64 * (2 * half + e), (int)g.M - 1);
        cb[e] = min(lane + 64 * (2 * half + e), g.N - 1);
    }
    const bool do_cs = g.colsum != nullptr && !seg2;
    float cs[2] = {0.f, 0.f};
    auto fetch = [&](int t, float (&va)[2][4], float (&vb)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kc = min(k_begin + t * XK + 4 * kb + j, k_end - 1);  // wave-uniform row
            const float* arow = Ap + (int64_t)kc * lda;
            const float* brow = Bp + (int64_t)kc * ldb;
#pragma unroll
            for (int e = 0; e < 2; ++e) { va[e][j] = arow[ca[e]]; vb[e][j] = brow[cb[e]]; }
        }
    };
    auto stage = [&](int t, int buf, float (&va)[2][4], float (&vb)[2][4]) __attribute__((always_inline)) {
        // rows behind k_end are clamped duplicates: zeroed here (both operands: 0 x finite)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kin = k_begin + t * XK + 4 * kb + j < k_end;
#pragma unroll
            for (int e = 0; e < 2; ++e) { va[e][j] = kin ? va[e][j] : 0.f; vb[e][j] = kin ? vb[e][j] : 0.f; }
        }
        // (always summed, stored only when asked for: a branch here would cut the step's scheduling region in two)
#pragma unroll
        for (int e = 0; e < 2; ++e) cs[e] += (va[e][0] + va[e][1]) + (va[e][2] + va[e][3]);
        unsigned char* base = ldsx + buf * X_BUF;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int h0, m0, l0, h1, m1, l1;
            x3g_split2(va[e][0], va[e][1], h0, m0, l0);
            x3g_split2(va[e][2], va[e][3], h1, m1, l1);
            unsigned char* d = base + (lane + 64 * (2 * half + e)) * X_ROW + kb * 8;
            *reinterpret_cast<int2*>(d) = make_int2(h0, h1);
            if constexpr (NPL >= 2) *reinterpret_cast<int2*>(d + X_PLANE) = make_int2(m0, m1);
            if constexpr (NPL >= 3) *reinterpret_cast<int2*>(d + 2 * X_PLANE) = make_int2(l0, l1);
            x3g_split2(vb[e][0], vb[e][1], h0, m0, l0);
            x3g_split2(vb[e][2], vb[e][3], h1, m1, l1);
            d += X_OPND;
            *reinterpret_cast<int2*>(d) = make_int2(h0, h1);
            if constexpr (NPL >= 2) *reinterpret_cast<int2*>(d + X_PLANE) = make_int2(m0, m1);
            if constexpr (NPL >= 3) *reinterpret_cast<int2*>(d + 2 * X_PLANE) = make_int2(l0, l1);
        }
    };
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* abase = ldsx + buf * X_BUF + (wr * 128 + li) * X_ROW + lh * 16;
        const unsigned char* bbase = ldsx + buf * X_BUF + X_OPND + (wc * 64 + li) * X_ROW + lh * 16;
        gbf16x8 fa[4][NPL];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa[i][p] = *reinterpret_cast<const gbf16x8*>(abase + i * 32 * X_ROW + p * X_PLANE);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            gbf16x8 fb[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) fb[p] = *reinterpret_cast<const gbf16x8*>(bbase + n * 32 * X_ROW + p * X_PLANE);
            // the partial products, smallest first; consecutive MFMAs go to different accumulators
            if constexpr (NP == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[0], acc[i][n], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[2], acc[i][n], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[1], acc[i][n], 0, 0, 0);
            }
            if constexpr (NP >= 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[0], acc[i][n], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[1], acc[i][n], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[0], acc[i][n], 0, 0, 0);
        }
    };
    // prologue: tile 0 into buffer 0, tiles 1 and 2 in flight
    fetch(0, ra[0], rb[0]);
    stage(0, 0, ra[0], rb[0]);
    fetch(1, ra[0], rb[0]);
    fetch(2, ra[1], rb[1]);
    __syncthreads();
    // step t: multiply tile t (buffer t & 1); split tile t + 1 (register set t & 1) into the other buffer; request tile t + 3 into that set
    // (measured: the two waves of a SIMD running the halves of a step in OPPOSITE order -- one multiplies while the other splits --
    //  is slower, 6.27 against 5.66 ms; the time of a step is the SUM of its matrix and its vector / memory work however they are
    //  arranged, 2.45 ms + 0.67 ms per partial product at 8.6 G rows x columns: the chip is power-bound on this kernel, so an
    //  arrangement changes nothing and only fewer joules per product would)
#define X3_STEP(T, S)                                       \
    {                                                       \
        if (X3_DO_MFMA) multiply((T) & 1);                  \
        if (X3_DO_STAGE) stage((T) + 1, ((T) + 1) & 1, ra[S], rb[S]); \
        if (X3_DO_FETCH) fetch((T) + 3, ra[S], rb[S]);      \
        lds_barrier();                                      \
    }
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        X3_STEP(t, 0)
        X3_STEP(t + 1, 1)
    }
    if (t < nt) X3_STEP(t, 0)
#undef X3_STEP
    __syncthreads();
    if (do_cs) {  // thread (kb, half, lane) summed columns lane + 64 (2 half + e) over its k rows: reduce over the 4 k-row groups
        float* red = reinterpret_cast<float*>(ldsx);
#pragma unroll
        for (int e = 0; e < 2; ++e) red[kb * 256 + lane + 64 * (2 * half + e)] = cs[e];
        __syncthreads();
        if (tid < 256) {
            const float v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
            if (tid < g.M) g.colsum[(int64_t)split * g.M + tid] = v;
        }
    }
    // partial tile -> workspace.  Lane (li, lh) holds C[wr*128 + i*32 + (r&3) + 8*(r>>2) + 4*lh][wc*64 + n*32 + li].
    float* Cw = g.C + (int64_t)split * g.split_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = wc * 64 + n * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) Cw[(int64_t)row * g.N + col] = acc[i][n][r];
            }
        }
}

// ---- weight gradients of an input block: 256 x (<= 64) outputs, ONE tile per workgroup ------------------------------
// dW_in = dZ^T X_in with a narrow X_in (the 39 / 33 encoding columns that enter layer 0 of the stage-1 networks,
// stage1/model/network.py:85-106): HBM-bound -- per k-row 1 KB of dZ against 2 x 256 x 64 MACs -- so the job is to read
// dZ exactly once and to keep enough rows in flight, not to keep the matrix pipe busy.  The 128 x 128 tiles of
// gemm_tn_grouped_kernel pad N to 128 (3x the MFMA work) and split M over two workgroups that both read X_in:
// ~40 TFLOP/s.  Here: 4 waves x (64 x 64) = 2 x 2 MFMA tiles, 21 KB of LDS per buffer, three workgroups per CU, two
// k-tiles of operand rows in flight per workgroup (same staging scheme as the 256 x 256 kernel).
constexpr int T64 = 64, T64_LD = T64 + 4, TALL_BUF = BK * (T256_LD + T64_LD);
__device__ __forceinline__ void fetch_tile64(const float* __restrict__ src, int64_t ld, int n_cols, int k0, int k_end, int tid, float4& v) {
    const int cq = min((tid & 15) * 4, ((n_cols - 1) >> 2) << 2);
    const int kc = min(k0 + (tid >> 4), k_end - 1);
    v = *reinterpret_cast<const float4*>(src + (int64_t)kc * ld + cq);
}
__device__ __forceinline__ void mask_tile64(int n_cols, int k0, int k_end, int tid, float4& v) {
    const int cq = (tid & 15) * 4;
    const bool kin = k0 + (tid >> 4) < k_end;
    v.x = (kin && cq + 0 < n_cols) ? v.x : 0.f;
    v.y = (kin && cq + 1 < n_cols) ? v.y : 0.f;
    v.z = (kin && cq + 2 < n_cols) ? v.z : 0.f;
    v.w = (kin && cq + 3 < n_cols) ? v.w : 0.f;
}
__global__ __launch_bounds__(256, 3) void gemm_tn_tall_grouped_kernel(GroupedArgs gg) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TALL_BUF];  // 2 x {A[16][260], B[16][68]}
    int gi = 0;
    while (gi + 1 < gg.n && (int64_t)blockIdx.x >= gg.block_start[gi + 1]) ++gi;
    const GemmArgs g = gg.g[gi];
    const int split = (int)((int64_t)blockIdx.x - gg.block_start[gi]);
    if (split >= g.split_k) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const bool seg2 = split >= g.seg_splits;
    const float* Ap = seg2 ? g.A2 : g.A;
    const float* Bp = seg2 ? g.B2 : g.B;
    const int64_t lda = seg2 ? g.lda2 : g.lda, ldb = seg2 ? g.ldb2 : g.ldb;
    const int k_begin = (seg2 ? split - g.seg_splits : split) * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int nt = (k_end - k_begin + BK - 1) / BK;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
    float4 ra[2][4], rb[2];
    const bool do_cs = g.colsum != nullptr && !seg2;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define TL_FETCH(T, S)                                                                          \
    fetch_tile256(Ap, lda, (int)g.M, k_begin + (T) * BK, k_end, tid, ra[S]);                    \
    fetch_tile64(Bp, ldb, g.N, k_begin + (T) * BK, k_end, tid, rb[S]);
#define TL_STORE(BUF, S, T)                                                                     \
    mask_tile256((int)g.M, k_begin + (T) * BK, k_end, tid, ra[S]);                              \
    mask_tile64(g.N, k_begin + (T) * BK, k_end, tid, rb[S]);                                    \
    if (do_cs) {                                                                                \
        cs.x += (ra[S][0].x + ra[S][1].x) + (ra[S][2].x + ra[S][3].x); cs.y += (ra[S][0].y + ra[S][1].y) + (ra[S][2].y + ra[S][3].y); \
        cs.z += (ra[S][0].z + ra[S][1].z) + (ra[S][2].z + ra[S][3].z); cs.w += (ra[S][0].w + ra[S][1].w) + (ra[S][2].w + ra[S][3].w); \
    }                                                                                           \
    store_tile<false, T256>(lds + (BUF) * TALL_BUF, tid, ra[S]);                                \
    *reinterpret_cast<float4*>(lds + (BUF) * TALL_BUF + BK * T256_LD + (tid >> 4) * T64_LD + (tid & 15) * 4) = rb[S];
    // step T: multiply tile T out of buffer T & 1; register set S = T & 1 holds tile T + 1 (requested two steps ago): write
    // it to the other buffer and re-use the set for tile T + 3.  No branches around the staging (addresses are clamped,
    // rows beyond the slice are masked to zero): with the loads of tiles T + 2 and T + 3 in ONE basic block the compiler
    // counts its vmcnt waits instead of draining the queue, i.e. two tiles really stay in flight.  MFMA operands of k-pair
    // j + 1 are read from LDS while the MFMAs of k-pair j execute (two register sets, order pinned).
#define TL_STEP(T, S)                                                                           \
    {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        const float* As = lds + ((T) & 1) * TALL_BUF + wave * 64 + li;                          \
        const float* Bs = lds + ((T) & 1) * TALL_BUF + BK * T256_LD + li;                       \
        float pa[2][2], pb[2][2];                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) pa[0][i] = As[lh * T256_LD + i * 32];     \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) pb[0][n] = Bs[lh * T64_LD + n * 32];      \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                         \
            if (j < 7) {                                                                        \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) pa[(j + 1) & 1][i] = As[(2 * j + 2 + lh) * T256_LD + i * 32]; \
                _Pragma("unroll") for (int n = 0; n < 2; ++n) pb[(j + 1) & 1][n] = Bs[(2 * j + 2 + lh) * T64_LD + n * 32];  \
            }                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                       \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                   \
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j & 1][i], pb[j & 1][n], acc[i][n], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                  \
        }                                                                                       \
        TL_STORE(((T) + 1) & 1, S, (T) + 1)                                                     \
        TL_FETCH((T) + 3, S)                                                                    \
        lds_barrier(); /* orders LDS only: the rows of tiles T + 2, T + 3 stay in flight */     \
    }
    TL_FETCH(0, 0)
    TL_STORE(0, 0, 0)
    TL_FETCH(1, 0)
    TL_FETCH(2, 1)
    __syncthreads();
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        TL_STEP(t, 0)
        TL_STEP(t + 1, 1)
    }
    if (t < nt) TL_STEP(t, 0)
#undef TL_STEP
#undef TL_STORE
#undef TL_FETCH
    if (do_cs) {  // thread tid staged columns 4 (tid % 64) .. +3 (rows tid / 64 + 4u of every k-tile): reduce over the 4 waves
        float4* red = reinterpret_cast<float4*>(lds);
        red[wave * 64 + lane] = cs;
        __syncthreads();
        if (tid < 64) {
            const float4 a = red[tid], b = red[64 + tid], c = red[128 + tid], d = red[192 + tid];
            const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
            float* dst = g.colsum + (int64_t)split * g.M + 4 * tid;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * tid + e < g.M) dst[e] = v[e];
        }
    }
    // partial tile -> workspace.  Lane (li, lh) holds C[wave*64 + i*32 + (r&3) + 8*(r>>2) + 4*lh][n*32 + li].
    float* Cw = g.C + (int64_t)split * g.split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = n * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) Cw[(int64_t)row * g.N + col] = acc[i][n][r];
            }
        }
}

struct ReduceItem {
    const float* ws;       // [splits][M*N] partial outputs
    const float* cs_ws;    // [cs_splits][M] partial column sums or nullptr
    float* C;
    float* colsum;
    int64_t MN, ldc;
    int N, M, splits, cs_splits, accumulate, vec;
    int blocks;            // blocks that reduce C; the following ceil(M/256) blocks reduce the column sums
};
struct GroupedReduceArgs {
    ReduceItem it[kMaxGroup];
};
__global__ __launch_bounds__(256) void grouped_reduce_kernel(GroupedReduceArgs ra) {
    const ReduceItem r = ra.it[blockIdx.y];
    __shared__ float4 red[8][32];
    const int b = blockIdx.x;
    if (b < r.blocks) {
        if (r.vec) {  // same scheme as splitk_reduce_vec_kernel
            const int tx = threadIdx.x & 31, tz = threadIdx.x >> 5;
            const int64_t i = ((int64_t)b * 32 + tx) * 4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < r.MN) {
                const int per = (r.splits + 7) / 8;
                const int z0 = tz * per, z1 = min(r.splits, z0 + per);
#pragma unroll 8
                for (int z = z0; z < z1; ++z) {
                    const float4 v = *reinterpret_cast<const float4*>(r.ws + (int64_t)z * r.MN + i);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            red[tz][tx] = s;
            __syncthreads();
            if (tz == 0 && i < r.MN) {
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const float4 v = red[k][tx];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                const int64_t m = i / r.N, n = i % r.N;
                float4* cp = reinterpret_cast<float4*>(r.C + m * r.ldc + n);
                if (r.accumulate) {
                    const float4 c = *cp;
                    s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
                }
                *cp = s;
            }
        } else {
            const int64_t i = (int64_t)b * 256 + threadIdx.x;
            if (i < r.MN) {
                float s = 0.0f;
                for (int z = 0; z < r.splits; ++z) s += r.ws[(int64_t)z * r.MN + i];
                const int64_t m = i / r.N, n = i % r.N;
                float* cp = r.C + m * r.ldc + n;
                *cp = r.accumulate ? (*cp + s) : s;
            }
        }
    } else if (r.colsum != nullptr) {
        const int64_t m = (int64_t)(b - r.blocks) * 256 + threadIdx.x;
        if (m < r.M) {
            float s = 0.0f;
            for (int z = 0; z < r.cs_splits; ++z) s += r.cs_ws[(int64_t)z * r.M + m];
            r.colsum[m] = s;
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int64_t MN, int N,
                                                            int64_t ldc, int splits, int accumulate,
                                                            float* __restrict__ C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < MN; i += (int64_t)gridDim.x * 256) {
        float s = 0.0f;
        for (int z = 0; z < splits; ++z) s += ws[(int64_t)z * MN + i];
        int64_t m = i / N, n = i % N;
        float* cp = C + m * ldc + n;
        *cp = accumulate ? (*cp + s) : s;
    }
}

// Same sum for the common aligned case (N, ldc multiples of 4): a block owns 128 consecutive outputs as 32 float4
// columns x 8 slices of the split range, so a 256x256 gradient with 128 partials is reduced by 512 blocks with 16
// independent 16-byte loads per thread instead of 128 scalar ones.  Fixed summation order (slice-major): deterministic.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ ws, int64_t MN, int N,
                                                                int64_t ldc, int splits, int accumulate,
                                                                float* __restrict__ C) {
    __shared__ float4 red[8][32];
    const int tx = threadIdx.x & 31, tz = threadIdx.x >> 5;
    const int64_t i = ((int64_t)blockIdx.x * 32 + tx) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < MN) {
        const int per = (splits + 7) / 8;
        const int z0 = tz * per, z1 = min(splits, z0 + per);
#pragma unroll 8
        for (int z = z0; z < z1; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)z * MN + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[tz][tx] = s;
    __syncthreads();
    if (tz == 0 && i < MN) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float4 v = red[k][tx];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int64_t m = i / N, n = i % N;  // N % 4 == 0: the four outputs share a row
        float4* cp = reinterpret_cast<float4*>(C + m * ldc + n);
        if (accumulate) {
            const float4 c = *cp;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        }
        *cp = s;
    }
}

// column sums, two stages: partial[b][n] = sum over a row slab, then out[n] = sum_b partial[b][n]
// with row weights w [M, n_w] (n_w <= 4): partial sums of w[m][j] * X[m][n] (product and sum rounded separately, as the
// elementwise product followed by a column sum would be): the n_w x N weight gradient g^T H of a layer with n_w <= 4
// outputs as a bandwidth-bound reduction (X is read once) instead of an n_w-row GEMM item that occupies a whole
// 128-row tile per K slice
template <int NW>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                             int64_t ldw, int64_t M, int N, int64_t ldx,
                                                             int64_t rows_per_block, float* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += 256) {
        if (NW == 0) {
            float s = 0.0f;
            for (int64_t m = r0; m < r1; ++m) s += X[m * ldx + n];
            partial[(int64_t)blockIdx.x * N + n] = s;
        } else {
            float s[NW > 0 ? NW : 1];
#pragma unroll
            for (int j = 0; j < NW; ++j) s[j] = 0.0f;
            for (int64_t m = r0; m < r1; ++m) {
                const float x = X[m * ldx + n];
#pragma unroll
                for (int j = 0; j < NW; ++j) s[j] += w[m * ldw + j] * x;
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) partial[((int64_t)blockIdx.x * NW + j) * N + n] = s[j];
        }
    }
}
// The same with 16-byte loads: a thread owns 4 consecutive columns and every fourth row of the slab (64 threads x 16 B =
// one 1 KB row piece per wave and load, four rows of a thread in flight through the unrolled loop); the four row phases
// are combined through LDS in a fixed order.  The one-column-per-thread kernel above is latency-bound (one dependent
// load -> multiply -> add chain per thread: 1.5 TB/s on [524288, 256]); this one streams.
template <int NW>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                                 int64_t ldw, int64_t M, int N, int64_t ldx,
                                                                 int64_t rows_per_block, float* __restrict__ partial) {
    constexpr int NW1 = NW > 0 ? NW : 1;
    __shared__ float4 red[3][NW1][64];
    const int c4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    float4 s[NW1];
#pragma unroll
    for (int j = 0; j < NW1; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < N) {
#pragma unroll 4
        for (int64_t m = r0 + rsub; m < r1; m += 4) {
            const float4 x = *reinterpret_cast<const float4*>(X + m * ldx + c4);
#pragma unroll
            for (int j = 0; j < NW1; ++j) {
                const float wj = NW > 0 ? w[m * ldw + j] : 1.0f;
                if (NW > 0) { s[j].x += wj * x.x; s[j].y += wj * x.y; s[j].z += wj * x.z; s[j].w += wj * x.w; }
                else { s[j].x += x.x; s[j].y += x.y; s[j].z += x.z; s[j].w += x.w; }
            }
        }
    }
    if (rsub > 0) {
#pragma unroll
        for (int j = 0; j < NW1; ++j) red[rsub - 1][j][threadIdx.x & 63] = s[j];
    }
    __syncthreads();
    if (rsub == 0 && c4 < N) {
#pragma unroll
        for (int j = 0; j < NW1; ++j) {
            const float4 a = red[0][j][threadIdx.x], b = red[1][j][threadIdx.x], c = red[2][j][threadIdx.x];
            float4 t;
            t.x = (s[j].x + a.x) + (b.x + c.x); t.y = (s[j].y + a.y) + (b.y + c.y);
            t.z = (s[j].z + a.z) + (b.z + c.z); t.w = (s[j].w + a.w) + (b.w + c.w);
            *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * NW1 + j) * N + c4) = t;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int N, int NW,
                                                           int accumulate, float* __restrict__ out) {
    __shared__ float red[4];
    const int j = blockIdx.x / N, n = blockIdx.x % N;  // out[j][n]
    float s = 0.0f;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[((int64_t)b * NW + j) * N + n];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        out[blockIdx.x] = accumulate ? out[blockIdx.x] + tot : tot;
    }
}

}  // namespace psn

extern "C" int psn_gemm(int trans_a, int trans_b, int64_t M, int N, int K, const float* A, int64_t lda,
                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue,
                        const float* aux_in, int64_t ld_aux_in, const float* aux_in2, int64_t ld_aux_in2,
                        float* aux_out, int64_t ld_aux_out, int split_k, float* workspace, float* colsum_a, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(A && B && C, "gemm: null operand");
    PSN_CHECK_ARG(colsum_a == nullptr || trans_a, "gemm: colsum_a needs trans_a (A stored [K][M])");
    PSN_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    PSN_CHECK_ARG(epilogue >= 0 && epilogue <= PSN_EPI_MUL_AUX_RAW, "gemm: unknown epilogue %d", epilogue);
    if (epilogue == PSN_EPI_MUL_AUX || epilogue == PSN_EPI_MUL_POS || epilogue >= PSN_EPI_MUL2) PSN_CHECK_ARG(aux_in, "gemm: epilogue needs aux_in");
    if (epilogue == PSN_EPI_MUL2 || epilogue == PSN_EPI_SOFTPLUS_BWD) PSN_CHECK_ARG(aux_in2, "gemm: epilogue needs aux_in2");
    if (epilogue == PSN_EPI_MUL2 || epilogue == PSN_EPI_MUL_AUX_RAW) PSN_CHECK_ARG(aux_out, "gemm: epilogue needs aux_out");
    if (epilogue >= PSN_EPI_BIAS && epilogue <= PSN_EPI_BIAS_SOFTPLUS) PSN_CHECK_ARG(bias, "gemm: epilogue needs bias");
    if (epilogue == PSN_EPI_BIAS_SIGMOID) PSN_CHECK_ARG(bias, "gemm: epilogue needs bias");
    if (split_k < 1) split_k = 1;
    if (split_k > 1) {
        PSN_CHECK_ARG(epilogue == PSN_EPI_NONE || epilogue == PSN_EPI_ACCUM, "gemm: split_k only with NONE/ACCUM epilogue");
        PSN_CHECK_ARG(workspace, "gemm: split_k needs workspace");
    }
    if (M == 0) return PSN_OK;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.epi = epilogue; g.aux_in = aux_in; g.ld_aux_in = ld_aux_in; g.aux_out = aux_out;
    g.aux_in2 = aux_in2; g.ld_aux_in2 = ld_aux_in2;
    g.ld_aux_out = ld_aux_out;
    int64_t tiles_m = (M + BM - 1) / BM;
    // 256-column tiles read each A row-panel once (measured +0-5 % on the Q x 256 x 256 layer GEMMs) but halve the
    // workgroup count; keep them only when the grid still covers the 256 CUs several times over.
    const int64_t tiles_m_ = (M + BM - 1) / BM;
    int bn = N > 128 ? 256 : 128;
    if (bn == 256 && tiles_m_ * ((N + 255) / 256) * (split_k < 1 ? 1 : split_k) < 2048) bn = 128;
    g.tiles_n = (N + bn - 1) / bn;
    g.n_tiles = tiles_m * g.tiles_n;
    PSN_CHECK_ARG(g.n_tiles < (1ll << 31), "gemm: too many tiles");
    g.a_vec = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0);
    g.b_vec = (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0);
    g.auxin_vec = aux_in && (((uintptr_t)aux_in & 15) == 0) && (ld_aux_in % 4 == 0);
    g.auxin2_vec = aux_in2 && (((uintptr_t)aux_in2 & 15) == 0) && (ld_aux_in2 % 4 == 0);
    g.auxout_vec = aux_out && (((uintptr_t)aux_out & 15) == 0) && (ld_aux_out % 4 == 0);
    int kc = (K + split_k - 1) / split_k;
    kc = ((kc + BK - 1) / BK) * BK;
    split_k = (K + kc - 1) / kc;
    g.k_chunk = kc;
    g.split_stride = 0;
    if (split_k > 1) {
        g.C = workspace;
        g.ldc = N;
        g.split_stride = M * (int64_t)N;
        g.epi = PSN_EPI_NONE;
    }
    g.c_vec = (((uintptr_t)g.C & 15) == 0) && (g.ldc % 4 == 0) && (g.split_stride % 4 == 0);
    g.split_k = split_k;
    // column sums of A: straight to the caller's buffer, or per-split partials behind the C partials
    g.colsum = colsum_a == nullptr ? nullptr : (split_k > 1 ? workspace + (int64_t)split_k * M * N : colsum_a);
    g.A2 = nullptr; g.B2 = nullptr; g.lda2 = g.ldb2 = 0; g.seg_splits = split_k; g.b_div = g.b_mod = 0; g.Bt2 = nullptr; g.ldbt2 = 0; g.b2_div = g.b2_mod = g.b_split = 0;
    PSN_CHECK_ARG(g.n_tiles * split_k < (1ll << 31), "gemm: too many blocks");
    dim3 grid((unsigned)(g.n_tiles * split_k)), block(256);
#define PSN_LAUNCH(TA_, TB_)                                                                          \
    if (bn == 256) hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 4>), grid, block, 0, st, g);             \
    else hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 2>), grid, block, 0, st, g);
    if (!trans_a && trans_b) { PSN_LAUNCH(false, true) }
    else if (!trans_a && !trans_b) { PSN_LAUNCH(false, false) }
    else if (trans_a && !trans_b) { PSN_LAUNCH(true, false) }
    else { PSN_LAUNCH(true, true) }
#undef PSN_LAUNCH
    PSN_CHECK_LAUNCH("gemm");
    if (split_k > 1) {
        int64_t MN = M * (int64_t)N;
        const int acc = epilogue == PSN_EPI_ACCUM ? 1 : 0;
        if (N % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)C | (uintptr_t)workspace) & 15) == 0 && MN / 128 < (1ll << 31)) {
            hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)((MN + 127) / 128)), dim3(256), 0, st, workspace, MN, N,
                               ldc, split_k, acc, C);
        } else {
            int64_t blocks = (MN + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, workspace, MN, N, ldc, split_k, acc, C);
        }
        PSN_CHECK_LAUNCH("gemm split-k reduce");
        if (colsum_a != nullptr) {  // [split_k][M] partial column sums -> colsum_a[M]
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, g.colsum, M, (int)M, M,
                               split_k, 0, colsum_a);
            PSN_CHECK_LAUNCH("gemm split-k colsum reduce");
        }
    }
    return PSN_OK;
}

static int gemm_tn_grouped_impl(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                                int64_t workspace_floats, void* stream, bool x3);
// partial products per multiply of psn_gemm_tn_grouped_x3: 6 (default: fp32-class), 3 (~16 significant bits), 1 (plain bf16 operands)
static int g_x3_products = 6;
extern "C" int psn_gemm_tn_x3_set_products(int n_products) {
    const int prev = g_x3_products;
    if (n_products == 6 || n_products == 3 || n_products == 1) g_x3_products = n_products;
    return prev;
}
extern "C" int psn_gemm_tn_grouped(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    return gemm_tn_grouped_impl(n_items, items, K, split_k, workspace, workspace_floats, stream, false);
}
extern "C" int psn_gemm_tn_grouped_x3(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                                      int64_t workspace_floats, void* stream) {
    return gemm_tn_grouped_impl(n_items, items, K, split_k, workspace, workspace_floats, stream, true);
}
static int gemm_tn_grouped_impl(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                                int64_t workspace_floats, void* stream, bool x3) {
    using namespace psn;
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= kMaxGroup, "gemm_tn_grouped: n_items=%d (1..%d)", n_items, kMaxGroup);
    PSN_CHECK_ARG(K > 0 && K < (1ll << 31) && workspace, "gemm_tn_grouped: bad K or null workspace");
    if (split_k < 1) split_k = 1;
    int kc = (int)((K + split_k - 1) / split_k);
    kc = ((kc + BK - 1) / BK) * BK;
    split_k = (int)((K + kc - 1) / kc);
    // Products with 128 < M, N <= 256 (the hidden-layer gradients) take the one-tile-per-workgroup kernel: exactly one
    // workgroup per (product, K slice), one workgroup per CU, so the slice count is chosen to fill 256 CUs once.
    int big_products = 0, tall_products = 0;
    for (int i = 0; i < n_items; ++i) {
        const PsnGemmTnItem& it = items[i];
        PSN_CHECK_ARG(it.A && it.B && it.C && it.M > 0 && it.N > 0, "gemm_tn_grouped: item %d has a null operand or empty shape", i);
        PSN_CHECK_ARG((it.A2 == nullptr) == (it.B2 == nullptr), "gemm_tn_grouped: item %d needs both A2 and B2", i);
        const bool vec = (((uintptr_t)it.A | (uintptr_t)it.B | (uintptr_t)it.A2 | (uintptr_t)it.B2) & 15) == 0 && it.lda % 4 == 0 &&
                         it.ldb % 4 == 0 && it.lda >= 4 && it.ldb >= 4 && (!it.A2 || (it.lda2 % 4 == 0 && it.ldb2 % 4 == 0 && it.lda2 >= 4 && it.ldb2 >= 4));
        if (vec && it.b_div == 0 && it.M > 128 && it.N > 128 && it.M <= T256 && it.N <= T256) big_products += it.A2 ? 2 : 1;
        if (vec && it.b_div == 0 && it.M > 128 && it.M <= T256 && it.N <= T64) tall_products += it.A2 ? 2 : 1;
        PSN_CHECK_ARG(it.b_div == 0 || (it.b_div > 0 && it.b_mod > 0 && !it.A2), "gemm_tn_grouped: item %d bad table mapping", i);
        PSN_CHECK_ARG(it.B_tab2 == nullptr || (it.b_div > 0 && it.b2_div > 0 && it.b2_mod > 0 && it.b_split > 0 && it.b_split % 4 == 0 &&
                                                it.b_split < it.N), "gemm_tn_grouped: item %d bad second table", i);
    }
    int split_big = 1, kc_big = 0;
    if (big_products > 0) {
        int64_t want = 256 / big_products;
        if (want < 1) want = 1;
        if (want > K / 256) want = K / 256 > 0 ? K / 256 : 1;
        kc_big = (int)((K + want - 1) / want);
        kc_big = ((kc_big + TK - 1) / TK) * TK;
        split_big = (int)((K + kc_big - 1) / kc_big);
    }
    // 256 x (<= 64) products (input-block gradients): HBM- / MFMA-co-limited, about one workgroup per CU in total
    int split_tall = 1, kc_tall = 0;
    if (tall_products > 0) {
        int64_t want = 256 / tall_products;  // measured 192 ... 1024: 256 - 512 are best (kernel + reduction of the partial tiles)
        if (want < 1) want = 1;
        if (want > K / 256) want = K / 256 > 0 ? K / 256 : 1;
        kc_tall = (int)((K + want - 1) / want);
        kc_tall = ((kc_tall + BK - 1) / BK) * BK;
        split_tall = (int)((K + kc_tall - 1) / kc_tall);
    }
    GroupedArgs gg, gb, gt;  // 128 x 128 tiles | 256 x 256 tiles | 256 x 64 tiles
    GroupedReduceArgs ra;
    gg.n = gb.n = gt.n = 0;
    int64_t blocks = 0, blocks_big = 0, blocks_tall = 0, ws_off = 0;
    int max_rblocks = 1;
    for (int i = 0; i < n_items; ++i) {
        const PsnGemmTnItem& it = items[i];
        const int n_seg = it.A2 ? 2 : 1;
        const bool vec = (((uintptr_t)it.A | (uintptr_t)it.B | (uintptr_t)it.A2 | (uintptr_t)it.B2) & 15) == 0 && it.lda % 4 == 0 &&
                         it.ldb % 4 == 0 && it.lda >= 4 && it.ldb >= 4 && (!it.A2 || (it.lda2 % 4 == 0 && it.ldb2 % 4 == 0 && it.lda2 >= 4 && it.ldb2 >= 4));
        const bool big = vec && it.b_div == 0 && it.M > 128 && it.N > 128 && it.M <= T256 && it.N <= T256;
        const bool tall = vec && it.b_div == 0 && it.M > 128 && it.M <= T256 && it.N <= T64;
        const int sk = big ? split_big : tall ? split_tall : split_k;
        GemmArgs& g = big ? gb.g[gb.n] : tall ? gt.g[gt.n] : gg.g[gg.n];
        PSN_CHECK_ARG(it.k_rows >= 0 && it.k_rows <= K, "gemm_tn_grouped: item %d k_rows=%lld of K=%lld", i, (long long)it.k_rows, (long long)K);
        g.M = it.M; g.N = it.N; g.K = (int)(it.k_rows > 0 ? it.k_rows : K); g.A = it.A; g.lda = it.lda; g.B = it.B; g.ldb = it.ldb;
        g.A2 = it.A2; g.lda2 = it.lda2; g.B2 = it.B2; g.ldb2 = it.ldb2;
        // a modulo that cannot wrap ((K - 1) / div < mod) is passed as 0 = identity
        g.b_div = (int)it.b_div; g.b_mod = (it.b_div > 0 && (K - 1) / it.b_div < it.b_mod) ? 0 : (int)it.b_mod;
        g.Bt2 = it.B_tab2; g.ldbt2 = it.ldb_tab2; g.b2_div = (int)it.b2_div; g.b_split = it.b_split;
        g.b2_mod = (it.b2_div > 0 && (K - 1) / it.b2_div < it.b2_mod) ? 0 : (int)it.b2_mod;
        g.bias = nullptr; g.epi = PSN_EPI_NONE; g.aux_in = g.aux_in2 = nullptr; g.aux_out = nullptr;
        g.ld_aux_in = g.ld_aux_in2 = g.ld_aux_out = 0;
        g.tiles_n = (big || tall) ? 1 : (it.N + 127) / 128;
        g.n_tiles = (big || tall) ? 1 : (int64_t)((it.M + BM - 1) / BM) * g.tiles_n;
        g.k_chunk = big ? kc_big : tall ? kc_tall : kc;
        g.split_k = sk * n_seg;
        g.seg_splits = sk;
        g.a_vec = (((uintptr_t)it.A & 15) == 0) && (it.lda % 4 == 0) && (!it.A2 || ((((uintptr_t)it.A2 & 15) == 0) && (it.lda2 % 4 == 0)));
        g.b_vec = (((uintptr_t)it.B & 15) == 0) && (it.ldb % 4 == 0) && (!it.B2 || ((((uintptr_t)it.B2 & 15) == 0) && (it.ldb2 % 4 == 0))) &&
                  (!it.B_tab2 || ((((uintptr_t)it.B_tab2 & 15) == 0) && (it.ldb_tab2 % 4 == 0)));
        g.auxin_vec = g.auxin2_vec = g.auxout_vec = 0;
        const int64_t MN = (int64_t)it.M * it.N;
        g.C = workspace + ws_off;
        g.ldc = it.N;
        g.split_stride = MN;
        g.c_vec = ((ws_off % 4) == 0) && (it.N % 4 == 0) && (MN % 4 == 0) && (((uintptr_t)workspace & 15) == 0);
        ws_off += MN * g.split_k;
        ws_off = (ws_off + 3) / 4 * 4;
        g.colsum = it.colsum_a ? workspace + ws_off : nullptr;
        ReduceItem& r = ra.it[i];
        r.ws = g.C; r.cs_ws = g.colsum; r.C = it.C; r.colsum = it.colsum_a; r.MN = MN; r.ldc = it.ldc; r.N = it.N; r.M = it.M;
        r.splits = g.split_k; r.cs_splits = sk; r.accumulate = it.accumulate ? 1 : 0;
        r.vec = g.c_vec && (it.ldc % 4 == 0) && (((uintptr_t)it.C & 15) == 0);
        r.blocks = (int)(r.vec ? (MN + 127) / 128 : (MN + 255) / 256);
        const int rb = r.blocks + (it.colsum_a ? (it.M + 255) / 256 : 0);
        if (rb > max_rblocks) max_rblocks = rb;
        if (it.colsum_a) ws_off += (int64_t)sk * it.M;
        ws_off = (ws_off + 3) / 4 * 4;
        if (big) {
            gb.block_start[gb.n++] = blocks_big;
            blocks_big += g.split_k;
        } else if (tall) {
            gt.block_start[gt.n++] = blocks_tall;
            blocks_tall += g.split_k;
        } else {
            gg.block_start[gg.n++] = blocks;
            blocks += (g.n_tiles * g.split_k + 7) / 8 * 8;
        }
    }
    gg.block_start[gg.n] = blocks;
    gb.block_start[gb.n] = blocks_big;
    gt.block_start[gt.n] = blocks_tall;
    PSN_CHECK_ARG(ws_off <= workspace_floats, "gemm_tn_grouped: workspace too small (%lld floats needed)", (long long)ws_off);
    PSN_CHECK_ARG(blocks < (1ll << 31) && blocks_big < (1ll << 31) && blocks_tall < (1ll << 31), "gemm_tn_grouped: too many blocks");
    hipStream_t st = (hipStream_t)stream;
    if (gb.n > 0) {
        if (x3 && g_x3_products == 6) hipLaunchKernelGGL(gemm_tn256_x3_grouped_kernel<6>, dim3((unsigned)blocks_big), dim3(512), 2 * X_BUF, st, gb);
        else if (x3 && g_x3_products == 3) hipLaunchKernelGGL(gemm_tn256_x3_grouped_kernel<3>, dim3((unsigned)blocks_big), dim3(512), 2 * X_BUF, st, gb);
        else if (x3) hipLaunchKernelGGL(gemm_tn256_x3_grouped_kernel<1>, dim3((unsigned)blocks_big), dim3(512), 2 * X_BUF, st, gb);
        else hipLaunchKernelGGL(gemm_tn256_grouped_kernel, dim3((unsigned)blocks_big), dim3(256), T256_BUF * 2 * sizeof(float), st, gb);
        PSN_CHECK_LAUNCH("gemm_tn_grouped (256 x 256 tiles)");
    }
    if (gt.n > 0) {
        hipLaunchKernelGGL(gemm_tn_tall_grouped_kernel, dim3((unsigned)blocks_tall), dim3(256), 0, st, gt);
        PSN_CHECK_LAUNCH("gemm_tn_grouped (256 x 64 tiles)");
    }
    if (gg.n > 0) {
        hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gg);
        PSN_CHECK_LAUNCH("gemm_tn_grouped");
    }
    hipLaunchKernelGGL(grouped_reduce_kernel, dim3((unsigned)max_rblocks, (unsigned)n_items), dim3(256), 0, st, ra);
    PSN_CHECK_LAUNCH("gemm_tn_grouped reduce");
    return PSN_OK;
}

extern "C" int psn_colsum(const float* X, const float* row_weight, int n_w, int64_t ldw, int64_t M, int N, int64_t ldx,
                          float* out, int accumulate, float* workspace, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(X && out && workspace, "colsum: null pointer");
    PSN_CHECK_ARG(N > 0 && M >= 0, "colsum: bad shape");
    PSN_CHECK_ARG(row_weight == nullptr ? n_w == 0 : (n_w >= 1 && n_w <= 4 && ldw >= n_w), "colsum: n_w=%d (1..4 with weights, 0 without)", n_w);
    hipStream_t st = (hipStream_t)stream;
    const int nw1 = n_w > 0 ? n_w : 1;
    int nblocks = (int)((M + 127) / 128);
    if (nblocks > 2048 / nw1) nblocks = 2048 / nw1;  // workspace: 2048 * N floats
    if (nblocks < 1) nblocks = 1;
    int64_t rpb = (M + nblocks - 1) / nblocks;
    if (rpb < 1) rpb = 1;
    const bool vec = N <= 256 && N % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)workspace) & 15) == 0;
    if (vec) {
        switch (n_w) {
            case 0: hipLaunchKernelGGL(colsum_partial_vec_kernel<0>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
            case 1: hipLaunchKernelGGL(colsum_partial_vec_kernel<1>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
            case 2: hipLaunchKernelGGL(colsum_partial_vec_kernel<2>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
            case 3: hipLaunchKernelGGL(colsum_partial_vec_kernel<3>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
            default: hipLaunchKernelGGL(colsum_partial_vec_kernel<4>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
        }
    } else
    switch (n_w) {
        case 0: hipLaunchKernelGGL(colsum_partial_kernel<0>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
        case 1: hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
        case 2: hipLaunchKernelGGL(colsum_partial_kernel<2>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
        case 3: hipLaunchKernelGGL(colsum_partial_kernel<3>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
        default: hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3(nblocks), dim3(256), 0, st, X, row_weight, ldw, M, N, ldx, rpb, workspace); break;
    }
    PSN_CHECK_LAUNCH("colsum partial");
    // the partial sums of weight column j form an [nblocks, N] matrix with row stride n_w * N: out [n_w, N]
    hipLaunchKernelGGL(colsum_final_kernel, dim3(N * nw1), dim3(256), 0, st, workspace, nblocks, N, nw1, accumulate, out);
    PSN_CHECK_LAUNCH("colsum final");
    return PSN_OK;
}
