// Fully fused 256-wide MLP engine: activations never leave registers.
//
// Lean variant (mlp_infer_kernel<false>) -- the no-grad network evaluations of the reference, optionally leaving the
// hidden activations of a row range behind for a later backward pass:
//   stage2 visibility_net over (L + V)*Ns rows  (stage2/model/renderer.py:191-200, 251-262; vis.detach() at :197)
//   stage1 occupancy-only queries               (stage1/model/rendering.py:456-462 march, :537-540 secant,
//                                                :394-399 light_visibility; network.py:124-125)
// Chain variant (mlp_infer_kernel<true>) -- the same pipeline with a per-layer activation PROGRAM (row-major
// operands, dumps, side outputs): forward-with-dumps, ReLU backward, the stage-1 gradient sweep
// (network.py:108-120 under create_graph=True) and the adjoints of both, see include/psnerf_hip.h PSN_ACT_*.
//
// Formulation: per layer OUT^T[features, points] = W[features, K] * ACT^T[K, points] on
// v_mfma_f32_16x16x4_f32.  One wave owns 16 points (the MFMA N dimension = lane & 15); a lane holds 64 of
// the 256 features of its point in registers, in exactly the MFMA C/D layout
//     feature(mt, r, g) = 16*mt + 4*g + r,    g = lane >> 4,  r = 0..3,  mt = 0..15.
// Because K may be visited in any order as long as A and B agree, the D registers of layer l are fed
// straight back as the B operand of layer l+1 (k-step (kt, r) consumes register act[kt][r], whose four
// lane groups hold features 16kt + 4g + r); the weights are pre-packed to match (psn_mlp_pack_layer), so
// there is no transpose, no LDS round trip and no HBM traffic for activations.
//
// Workgroup = 4 waves = 64 points; two workgroups share a CU (2 waves per SIMD, <= 256 VGPRs each): while one
// wave of a SIMD is parked at the per-stage barrier, issuing its LDS-DMA or running its activation program, the
// other keeps the matrix pipe busy (the first 32x32x2 / 1-wave-per-SIMD version measured 73 % MFMA-busy; an
// 8-wave workgroup per CU is no faster: the stage barrier then couples all eight waves).
// Weights (L2-resident, <= 2.6 MB) stream through LDS in 32 KB stages (32 input features x 256 outputs)
// by LDS-DMA (global_load_lds_dwordx4), double buffered, shared by the 4 waves; biases sit in LDS.
//
// Roofline: MFMA-bound.  Per stage and wave: 128 MFMAs (4096 cycles) vs 32 ds_read_b128.  Measured (lean variant,
// visibility net): 0.96-0.97 of the dense fp32-MFMA peak algorithmic, matrix pipe 88 % busy at 2.30 GHz.
#include "common.h"
#include <stdlib.h>
#include <atomic>

namespace psn {

struct InferArgs {
    PsnMlpDesc d;
    const float* w;
    const float* b;
    const float* ta;
    int64_t a_div, a_mod;
    const float* tb;
    int64_t b_div, b_mod;
    const float* init_a;
    const float* init_b;
    int64_t n_rows;
    float* out;
    int n_bias;  // floats in the packed bias buffer (copied to LDS once per workgroup)
    float* save[PSN_MLP_MAX_LAYERS];  // per hidden layer: row-major [n_rows - save_row0, 256] activation dump, or nullptr
    int64_t save_row0;                // rows >= save_row0 are dumped (training rows ride along with inference rows)
    const float* mask[PSN_MLP_MAX_LAYERS];  // aux1: row-major [n_rows, 256] tensor read by the layer's activation
    const float* aux2[PSN_MLP_MAX_LAYERS];  // aux2: second row-major operand (PSN_ACT_MUL2 / PSN_ACT_SOFTPLUS_BWD)
    float* save2[PSN_MLP_MAX_LAYERS];       // second dump (sigmoid of PSN_ACT_SOFTPLUS100, raw acc of MUL_AUX, acc*aux2 of MUL2)
    const float* act_init;                  // optional row-major [act_init_rows, 256] initial activations (chains that start from a tensor)
    int64_t act_init_rows;                  // rows >= act_init_rows start from zero activations (a tensor that covers a row prefix only)
    // rank-k init (k <= 4): the layers with init_off >= 0 additionally start from sum_c rk_coef[row, c] * rk_basis[c, init_off + f]
    // -- the init table of a backward chain, d h = g_out W_last with 1..3 outputs, formed in registers instead of being
    // written to HBM by a K = 3 GEMM / a broadcast product and read back (1 KB per row each way)
    // dump tile masks (chain variant): bit mt set = the 16-feature tile mt of the layer's first / second dump is written.
    // A dump of which only a column range is ever read (the raw sweep values of the skip layer and of layer 0: 39 of 256
    // columns each, 1 GB per 524k points) is written for those tiles only.
    uint32_t save_tiles[PSN_MLP_MAX_LAYERS], save2_tiles[PSN_MLP_MAX_LAYERS];
    const float* rk_coef;   // [n_rows, rk_k]
    const float* rk_basis;  // [rk_k, init_stride]
    int rk_k;
    // SRC == 1 (fused secant root finder, psn_root_find): the rows are rays, the network input is computed in the kernel
    const long long* n_rows_dev;  // SRC == 2, optional: the row count lives on the device (n_rows = capacity = grid size)
    const int64_t* out_rows;      // SRC == 2, optional: row r's outputs go to out[out_rows[r] * n_out ...] (scatter)
    const float* ray_o;      // [n_rows, 3]
    const float* ray_d;      // [n_rows, 3]
    const float* bracket;    // [4, n_rows]: d_low, d_high, f_low, f_high
    float tau, pe_scale;
    int n_iter, pe_octaves;
    // SRC == 3 (ray-march sweep, psn_march_sweep): the rows are (ray, step) pairs, the query point is generated in the kernel
    const float* far;        // [n_rays] sphere exit depth
    const float* u;          // [n_steps] linspace(0, 1, n_steps)
    const float* omu;        // [n_steps] 1 - u
    float near;
    int n_steps;
    int* skip;               // [n_rays] or nullptr: INT_MAX - b = block b of this ray holds a sign change (the lowest such b wins), 0 = none
    unsigned long long* n_blocks;  // optional: counts the 64-step blocks that were evaluated (measurement only)
    // SRC == 0, lean variant, optional (psn_mlp_infer_padded): the rows in front of save_row0 come in groups of live_period
    // (one group per light), of which only the first live_count[0] (a device-side float: psn_surface_index's count) are real;
    // the all-padding 64-row blocks are not evaluated (zeros)
    const float* live_count;
    int64_t live_period;
    // sign bits of the dumped activations (lean variant, 256-wide): save_bits[l] [n_rows - save_row0, 4] uint64 -- the word of
    // (row, lane group g) holds bit 4 mt + r = (activation feature 16 mt + 4 g + r > 0); read back by PSN_ACT_RELU_BITS chains
    // through mask[l], 32 bytes per row and layer instead of the 1 KB activation row
    unsigned long long* save_bits[PSN_MLP_MAX_LAYERS];
    int tb_lds;  // lean variant: 1 = a workgroup whose rows share one B-table row reads that init row through LDS (A/B: PSN_TB_LDS=0)    // lean variant, SRC == 0, pair row sets row = group * pm_period + point (stage 2: group = light, renderer.py:163,183-193): the
    // workgroups visit the 64-row blocks POINT-TILE-major with the group (light) running fastest, every XCD a contiguous share of
    // that order -- the workgroups resident at one time then share a handful of point tiles, whose A-side init rows (U[Ns, 512],
    // 60 MB at 29k points) stay in the XCD's L2 instead of being streamed from the fabric once per light.  0 = row order.
    int64_t pm_period;
    int pm_groups;
};

constexpr int kStageFloats = 8192;  // 32 input features x 256 outputs = 32 KB
constexpr int kWaves = 4;  // per workgroup; two workgroups share a CU (2 waves per SIMD, out of phase)

// One 32-feature k-tile (two 16-feature register tiles b0, b1 = 8 MFMA k-steps) against NMT output tiles of 16.
// The weight fragments of a group of 8 output tiles (8 ds_read_b128 = 8 KB per wave) feed 32 MFMAs.  Two register
// sets: the reads of group g+1 are issued at the START of group g's MFMA stream (1024 MFMA cycles ahead).  With a
// single set hipcc can only issue them behind the last six MFMAs (192 cycles), which does not cover the LDS queueing
// when the eight waves of a CU, barrier-aligned, all ask for their 8 KB at the same moment (PMC: matrix pipe 83 % busy).
// `request_next(j)` issues LDS-DMA piece j of the next weight stage.  The pieces are requested BEHIND the first group's
// fragment reads and spread over its first MFMAs (sched_group_barrier): issued as a block ahead of the first MFMA their
// ~50 issue slots were the largest single loss of the kernel (8 % when measured by removing the stream); a piece's
// issue fits into the 32-cycle gap of an MFMA.
template <int NMT, int NACC, bool ASMDMA, typename RequestNext>
__device__ __forceinline__ void stage_compute(floatx4 (&acc)[NACC], const floatx4& b0, const floatx4& b1,
                                              const float4* __restrict__ wl, int lane, RequestNext request_next) {
    if constexpr (NMT >= 8) {
        constexpr int GE = NMT / 8;      // groups of 8 output tiles per 16-feature half of the k-tile
        constexpr int NG = 2 * GE;
        float4 a[2][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[0][m] = wl[m * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);  // the first group's reads are in flight (exposed once per stage)
        constexpr int NPW = NMT / 2;        // LDS-DMA pieces of the next stage per wave: spread over the first MFMAs below
        if constexpr (ASMDMA) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int e = g / GE, m0 = (g % GE) * 8;
                const floatx4& bs = e ? b1 : b0;
                const float4(&ag)[8] = a[g & 1];
                // The order is pinned by scheduling regions (the LDS-DMA pieces are asm statements, which no scheduling
                // group matches): four chunks of [2 prefetch reads of group g+1 | 2 MFMAs | NPW/4 pieces (g == 0)], then the
                // other 24 MFMAs of the group.
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (g < NG - 1) {
                        const int e1 = (g + 1) / GE, m1 = ((g + 1) % GE) * 8;
                        a[(g + 1) & 1][2 * k] = wl[(e1 * NMT + m1 + 2 * k) * 64 + lane];
                        a[(g + 1) & 1][2 * k + 1] = wl[(e1 * NMT + m1 + 2 * k + 1) * 64 + lane];
                    }
                    acc[m0 + 2 * k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[2 * k].x, bs[0], acc[m0 + 2 * k], 0, 0, 0);
                    acc[m0 + 2 * k + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[2 * k + 1].x, bs[0], acc[m0 + 2 * k + 1], 0, 0, 0);
                    if (g == 0) {
#pragma unroll
                        for (int j = 0; j < NPW / 4; ++j) request_next(k * (NPW / 4) + j);
                    }
                    if (g == 0 || g < NG - 1) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].y, bs[1], acc[m0 + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].z, bs[2], acc[m0 + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].w, bs[3], acc[m0 + m], 0, 0, 0);
                // the last group stays open: what follows the stage (the operand loads of a chain layer's activation program,
                // the next stage's waits) may be hoisted under its MFMAs
                if (g < NG - 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPW; ++j) request_next(j);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int e = g / GE, m0 = (g % GE) * 8;
                const floatx4& bs = e ? b1 : b0;
                if (g < NG - 1) {
                    const int e1 = (g + 1) / GE, m1 = ((g + 1) % GE) * 8;
#pragma unroll
                    for (int m = 0; m < 8; ++m) a[(g + 1) & 1][m] = wl[(e1 * NMT + m1 + m) * 64 + lane];
                }
                const float4(&ag)[8] = a[g & 1];
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].x, bs[0], acc[m0 + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].y, bs[1], acc[m0 + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].z, bs[2], acc[m0 + m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[m].w, bs[3], acc[m0 + m], 0, 0, 0);
                // pin the order: this group's 8 prefetch reads spread over its first 8 MFMAs (with the builtin LDS-DMA pieces,
                // which the VMEM scheduling group matches, in group 0), then the other 24 MFMAs
                if (g < NG - 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        if (g == 0) __builtin_amdgcn_sched_group_barrier(0x020, NPW / 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NMT / 2; ++j) request_next(j);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const floatx4& bs = e ? b1 : b0;
            float4 a[NMT];
#pragma unroll
            for (int m = 0; m < NMT; ++m) a[m] = wl[(e * NMT + m) * 64 + lane];
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, bs[0], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, bs[1], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, bs[2], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, bs[3], acc[m], 0, 0, 0);
        }
    }
}

// ---- split-bf16 form of a stage (experiment, PsnMlpDesc.w_format = PSN_W_BF16X2; the 256-wide networks: chain launches and the
// gradient-free lean / occupancy / sweep launches) ----
// The stage holds the SAME 32-feature k-tile in the same [2][NMT][64 lanes][16 B] geometry, but the two halves are PLANES instead of
// k-halves: plane 0 = the bf16 heads of the lane's eight weights W[16 mt + i][32 kt + 16 (s / 4) + 4 g + s % 4], s = 0..7, plane 1 =
// the bf16 of what the heads left (x = hi + mid + O(2^-16 x)).  v_mfma_f32_16x16x32_bf16 contracts 32 k per instruction with lane
// group g supplying k-slots 8 g .. 8 g + 7 of both operands, and the slots above are exactly the eight activations the lane holds of
// this k-tile (b0[0..3], b1[0..3]): the B operand is split in registers, nothing moves between lanes, and the C layout -- hence
// every activation program, dump and epilogue -- is that of the fp32 instruction.  Three partial products per multiply
// (hi hi + hi mid + mid hi, fp32 accumulation): 48 MFMAs of 16 cycles per k-tile instead of 128 of 32.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef int cintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int chain_cvt2(float a, float b) {
    f32x2 f = {a, b};
    return __builtin_bit_cast(int, __builtin_convertvector(f, cbf16x2));  // v_cvt_pk_bf16_f32 (round to nearest even): a in the low half
}
__device__ __forceinline__ void chain_split(const floatx4& b0, const floatx4& b1, cbf16x8& hi, cbf16x8& mid) {
    cintx4 h, m;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = q < 2 ? b0[2 * q] : b1[2 * q - 4], b = q < 2 ? b0[2 * q + 1] : b1[2 * q - 3];
        h[q] = chain_cvt2(a, b);
        m[q] = chain_cvt2(a - __builtin_bit_cast(float, h[q] << 16), b - __builtin_bit_cast(float, h[q] & (int)0xFFFF0000));
    }
    hi = __builtin_bit_cast(cbf16x8, h);
    mid = __builtin_bit_cast(cbf16x8, m);
}
template <int NMT, int NACC, bool PIN, typename RequestNext>
__device__ __forceinline__ void stage_compute_x3(floatx4 (&acc)[NACC], const floatx4& b0, const floatx4& b1,
                                                 const float4* __restrict__ wl_, int lane, RequestNext request_next) {
    const cbf16x8* wl = reinterpret_cast<const cbf16x8*>(wl_);
    cbf16x8 bh, bm;
    chain_split(b0, b1, bh, bm);
    if constexpr (NMT >= 8) {
        constexpr int GE = NMT / 8;  // groups of 8 output tiles per plane
        constexpr int NG = 2 * GE;
        cbf16x8 a[2][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[0][m] = wl[m * 64 + lane];
#pragma unroll
        for (int j = 0; j < NMT / 2; ++j) request_next(j);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int p = g / GE, m0 = (g % GE) * 8;
            if (g < NG - 1) {
                const int p1 = (g + 1) / GE, m1 = ((g + 1) % GE) * 8;
#pragma unroll
                for (int m = 0; m < 8; ++m) a[(g + 1) & 1][m] = wl[(p1 * NMT + m1 + m) * 64 + lane];
            }
            const cbf16x8(&ag)[8] = a[g & 1];
            if (p == 0) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag[m], bm, acc[m0 + m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag[m], bh, acc[m0 + m], 0, 0, 0);
            // PIN (chain launches: builtin LDS-DMA pieces, which the VMEM scheduling group matches; measured chains -2.7 %, the lean
            // launch with its asm pieces +12 %, and with the fp32 stage's scheduling regions instead +-0: it is bound by the LDS reads): the order as in the fp32 stage -- the next group's 8 fragment reads (and, in group 0,
            // the LDS-DMA pieces of the next stage) spread over this group's first 8 MFMAs
            if constexpr (PIN) {
            if (g < NG - 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (g == 0) __builtin_amdgcn_sched_group_barrier(0x020, NMT / 8, 0);
                }
                if (p == 0) __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NMT / 2; ++j) request_next(j);
        cbf16x8 a[2][NMT];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int m = 0; m < NMT; ++m) a[p][m] = wl[(p * NMT + m) * 64 + lane];
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], bm, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][m], bh, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], bh, acc[m], 0, 0, 0);
    }
}

// LDS-DMA one stage (NBLK x 1 KB) of packed weights; wave w moves the contiguous blocks [w NBLK/4, (w+1) NBLK/4): four
// consecutive 1 KB pieces share one base (global address and M0) and differ in the instruction offset only
// (compile-time trip count, no kernarg reloads inside the stage: an s_load there forces s_waitcnt lgkmcnt(0), which
// also drains the LDS reads).
template <int NBLK, bool ASMDMA>
__device__ __forceinline__ void stage_piece(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane, int j) {
    constexpr int NPW = NBLK / kWaves;
    static_assert(NBLK % kWaves == 0, "stage size");
    const int grp = j >> 2;
    const float* base = gsrc + (wave * NPW + grp * 4) * 256;             // wave-uniform: SGPR pair
    const unsigned lds = lds_addr(lds_dst + (wave * NPW + grp * 4) * 256);  // wave-uniform: M0
    if constexpr (!ASMDMA) {
    auto gp = (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + (unsigned)(lane * 16));
    auto lp = (__attribute__((address_space(3))) void*)(lds_dst + (wave * NPW + grp * 4) * 256);
    switch (j & 3) {
        case 0: __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0); break;
    }
    } else {
    const unsigned voff = lane * 16;
    switch (j & 3) {
        case 0: lds_dma_16<0>(base, lds, voff); break;
        case 1: lds_dma_16<1024>(base, lds, voff); break;
        case 2: lds_dma_16<2048>(base, lds, voff); break;
        default: lds_dma_16<3072>(base, lds, voff); break;
    }
    }
}
template <int NBLK, bool ASMDMA>
__device__ __forceinline__ void stage_load(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < NBLK / kWaves; ++j) stage_piece<NBLK, ASMDMA>(gsrc, lds_dst, wave, lane, j);
}

__device__ __forceinline__ floatx4 ld4(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    floatx4 v;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    return v;
}
__device__ __forceinline__ void st4(float* p, const floatx4& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// Activation program of one chain layer (CHAIN variant).  The accumulators acc hold z; the result becomes the next
// layer's B operands (act).  Operand tiles are row-major [n_rows, 256]: p1 / p2 point at this lane's 4 floats of
// tile 0, tile mt is 16 floats further.  All operand loads of the layer are issued back to back (one exposed
// latency per layer instead of one per tile) and all dump stores are issued at the end, where they complete under
// the next layer's MFMAs.  p1 / p2 are non-null whenever CODE needs them (validated on the host).
template <int CODE, int NMT, bool FROMA = false>
__device__ __forceinline__ void chain_activation(floatx4 (&acc)[NMT], floatx4 (&act)[NMT], const float* __restrict__ p1,
                                                 const float* __restrict__ p2, float* __restrict__ d1, float* __restrict__ d2,
                                                 const uint32_t m1, const uint32_t m2, const unsigned long long sign_bits = 0) {
    // *_A codes (single-dump experiment): a1 is the dumped softplus OUTPUT a = softplus_100(z) of the forward layer instead of its
    // sigmoid; s = sigmoid(100 z) = 1 - exp(-100 a) (and 1 - s = exp(-100 a)) is re-formed here, so the value pass writes one
    // tensor per layer instead of two.  a < 0 cannot come from a softplus: such columns (the encoding half of the skip layer's
    // [a | pe] input tile) are clamped to s = 0 -- their products meet zero weight rows, but must stay finite.
    // (a separate INSTANTIATION of the kernel, FROMA: as further cases of the one kernel the three programs pushed the 256-register
    //  chain kernel into 92 spilled registers -- for every chain, not only theirs)
    constexpr bool kFromA = FROMA && (CODE == PSN_ACT_MUL_AUX || CODE == PSN_ACT_MUL2 || CODE == PSN_ACT_SOFTPLUS_BWD);
    constexpr bool kNeed1 = CODE == PSN_ACT_RELU_MASK || CODE == PSN_ACT_MUL_AUX || CODE == PSN_ACT_MUL2 || CODE == PSN_ACT_SOFTPLUS_BWD;
    constexpr bool kNeed2 = CODE == PSN_ACT_MUL2 || CODE == PSN_ACT_SOFTPLUS_BWD;
    constexpr bool kSecond = CODE == PSN_ACT_SOFTPLUS100 || CODE == PSN_ACT_MUL2 || CODE == PSN_ACT_MUL_AUX;
    floatx4 t1[NMT], t2[NMT];
    if constexpr (kNeed1) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) t1[mt] = ld4(p1 + mt * 16);
    }
    if constexpr (kNeed2) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) t2[mt] = ld4(p2 + mt * 16);
    }
    if constexpr (CODE == PSN_ACT_HEAD) {  // side output: dump z, the activations stay for the next layer
        if (d1 != nullptr) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                if ((m1 >> mt) & 1) st4(d1 + mt * 16, acc[mt]);
        }
        return;
    }
    // One 16-feature tile at a time (interleaving 16 softplus chains spills registers); each tile's results are
    // stored as soon as they exist, so operand and second-value registers die tile by tile.
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        floatx4 o, o2;
        if constexpr (CODE == PSN_ACT_SOFTPLUS100) {  // two packed pairs per tile (common.h)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 a, sg;
                softplus100_pair<true>(f32x2{acc[mt][2 * h], acc[mt][2 * h + 1]}, a, sg);
                o[2 * h] = a.x; o[2 * h + 1] = a.y; o2[2 * h] = sg.x; o2[2 * h + 1] = sg.y;
            }
        } else if constexpr (CODE == PSN_ACT_MUL_AUX || CODE == PSN_ACT_MUL2 || CODE == PSN_ACT_SOFTPLUS_BWD) {
            // packed pairs (same operations in the same order as the element-wise expressions, two values per instruction)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 z = {acc[mt][2 * h], acc[mt][2 * h + 1]};
                f32x2 a1 = {t1[mt][2 * h], t1[mt][2 * h + 1]};
                if constexpr (kFromA) {  // s = 1 - exp(-100 a), in place of the operand (no further live value)
                    const f32x2 t = a1 * (-100.0f * 1.44269502162933349609375f);
                    a1 = 1.0f - f32x2{__builtin_amdgcn_exp2f(fminf(t.x, 0.0f)), __builtin_amdgcn_exp2f(fminf(t.y, 0.0f))};
                }
                f32x2 a, b = z;
                if constexpr (CODE == PSN_ACT_MUL_AUX) a = z * a1;
                else if constexpr (CODE == PSN_ACT_MUL2) { a = z * a1; b = z * f32x2{t2[mt][2 * h], t2[mt][2 * h + 1]}; }
                else a = pk_fma(a1, z, ((1.0f - a1) * 100.0f) * f32x2{t2[mt][2 * h], t2[mt][2 * h + 1]});  // t1 z + 100 (1 - t1) t2
                o[2 * h] = a.x; o[2 * h + 1] = a.y; o2[2 * h] = b.x; o2[2 * h + 1] = b.y;
            }
        } else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = acc[mt][r];
            o2[r] = z;
            if constexpr (CODE == PSN_ACT_RELU) o[r] = relu1(z);
            else if constexpr (CODE == PSN_ACT_RELU_MASK) o[r] = t1[mt][r] > 0.0f ? z : 0.0f;
            else if constexpr (CODE == PSN_ACT_RELU_BITS)  // (compile-time bit position: one AND with a constant per element)
                o[r] = (((4 * mt + r) < 32 ? (unsigned)sign_bits : (unsigned)(sign_bits >> 32)) & (1u << ((4 * mt + r) & 31))) != 0u ? z : 0.0f;
            else o[r] = z;
        }
        act[mt] = o;
        if (d1 != nullptr && ((m1 >> mt) & 1)) st4(d1 + mt * 16, o);
        if constexpr (kSecond) {
            if (d2 != nullptr && ((m2 >> mt) & 1)) st4(d2 + mt * 16, o2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Wait until this wave's LDS-DMA pieces have landed while leaving `stores` (0, NMT or 2 NMT) younger dump stores in flight.
template <int NMT>
__device__ __forceinline__ void wait_for_weights(int stores) {
    if (stores == NMT + 1) {  // a dump and its sign-bit word
        if constexpr (NMT == 16) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
        else if constexpr (NMT == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else if (stores == 2 * NMT) {
        if constexpr (NMT == 16) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if constexpr (NMT == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (stores == NMT) {
        if constexpr (NMT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (NMT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// CHAIN = false: lean inference / forward-with-dump path (NONE / RELU / SOFTPLUS100 activations, one dump per layer).
// CHAIN = true : general per-layer activation programs with row-major operands and two dumps (training chains).
// SRC = 2: the rows are points [Q, 3] and the positional encoding is computed in the kernel (psn_mlp_infer_pe: the stage-1
// occupancy queries of the march sweep and the shadow rays).
// SRC = 0: input features from tables (everything above).  SRC = 1: the fused secant root finder of the stage-1 ray march
// (stage1/model/rendering.py:525-555): every row is a ray with a bracket [d_low, d_high] around the first free -> occupied
// crossing; the kernel iterates  p = o + d_pred dir -> positional encoding -> occupancy network -> regula-falsi update
// n_iter times WITHOUT leaving the launch (the reference, and round 1, paid one network launch + one encoding launch +
// one update launch per iteration, and a host synchronisation to compact the hit rays first).  A ray's iterations
// depend on nothing but the ray, so no inter-workgroup exchange is needed; the launch is latency-bound (one
// workgroup per 64 rays) and costs about one serial pass through the network per iteration.
// SRC = 3: the ray-march sweep of stage1/model/rendering.py:447-462 (psn_march_sweep): row = (ray, step), a workgroup = the
// 64 consecutive steps of ONE ray (block b of the ray), workgroups in BLOCK-MAJOR order (block 0 of every ray first).  The
// point ray0 + dir * (near (1 - u_m) + far u_m) is formed in the prologue (the expressions of sample_points_kernel, bit-identical)
// and encoded as for SRC = 2.  The reference's result depends only on the values up to the FIRST sign change of a ray
// (rendering.py:472-504): a block that contains one (or a ray that starts inside the object) raises the ray's flag, and the
// workgroups of the ray's later blocks -- dispatched thousands of workgroups later -- leave in their prologue.  The flag is a
// hint: only pairs inside one block are examined, whoever misses it evaluates values that nothing reads.
// TRIM: some layer reads one activation k-tile fewer than the width (a separate instantiation: the conditional last stage costs
// the untrimmed visibility launch 0.7 %).
// FROMA (chain variant): the MUL_AUX / MUL2 / SOFTPLUS_BWD programs take the dumped softplus OUTPUT as a1 and re-form the sigmoid
// (single-dump experiment, PSN_ACT_*_A on the host side).
// X3 (NMT = 16; chain launches, and the gradient-free lean launches SRC = 0 / 2 / 3): the weight stages hold two bf16 planes (PSN_W_BF16X2) and the matrix work runs as three bf16 partial
// products (stage_compute_x3); the final layer of a launch with n_out <= 32 stays fp32 (its block is packed PSN_W_F32).
template <bool CHAIN, int NMT, int SRC = 0, bool TRIM = false, bool FROMA = false, bool X3 = false>  // NMT = hidden width / 16: 16 (256-wide networks), 8 (128-wide) or 4 (64-wide)
__global__ __launch_bounds__(256, 2) void mlp_infer_kernel(InferArgs g) {
    static_assert(!X3 || (NMT == 16 && SRC != 1), "split-bf16 stages: 256-wide networks; the root finder stays fp32");
    constexpr int W = 16 * NMT;
    // LDS-DMA flavour (common.h lds_dma_16): asm pieces + explicit scheduling regions for the lean variant (exact lgkmcnt
    // waits: visibility launch +1.9 %, march sweep +4 %), the builtin + scheduling groups for the chain variant (its
    // F2 / B1 / B2 chains measured 1-2 % SLOWER with the regions, with either piece flavour)
    constexpr bool kAsmDma = !CHAIN && SRC != 1;  // (chain flavours: 2-4 % slower with asm pieces, re-measured with the V-row chain)  // (the root finder is latency-bound and at the register limit: builtin)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x 32 KB weight stages + all biases
    float* bias_lds = smem + 2 * kStageFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: keeps the LDS-DMA bases in SGPRs
    const int lj = lane & 15, lg = lane >> 4;
    int64_t blk_ = (int64_t)blockIdx.x;
    if constexpr (SRC == 0 && !CHAIN) {
        // psn_mlp_infer_padded: the rows [0, save_row0) are groups of live_period rows (64 | live_period, 64 | save_row0) of which
        // the first live_count[0] are real.  The hardware hands workgroup i to XCD i % 8, and the all-padding blocks sit at the END
        // of every group: skipped where they stand they all fall on the same XCDs while the others carry the full work (measured:
        // the launch no faster than without the skip).  So the grid enumerates the REAL blocks first -- row order: workgroup i
        // is real block i % rb of group i / rb, then come the blocks behind save_row0; point-major order: see below -- and the
        // surplus workgroups at the END of the grid, one per all-padding block, write that block's zeros and leave.
        if (g.live_count != nullptr) {
            const int64_t bpg = g.live_period / (kWaves * 16), groups = g.save_row0 / g.live_period;
            int64_t rb = ((int64_t)*g.live_count + kWaves * 16 - 1) / (kWaves * 16);
            rb = rb < bpg ? rb : bpg;
            const int64_t real = groups * rb, tail = (g.n_rows - g.save_row0 + kWaves * 16 - 1) / (kWaves * 16);
            int64_t first_surplus = real + tail;
            bool mapped = false;
            if (g.pm_period > 0) {
                // point-major: tile pt of every group that has one (pt < rb: the shading groups and the dumped groups behind
                // save_row0, pt >= rb: the dumped groups only, which are evaluated in full), the group running fastest; XCD x =
                // workgroups x, x + 8, ... takes the x-th eighth of that order
                const unsigned all = (unsigned)g.pm_groups, vg = all - (unsigned)groups;
                const unsigned btot = (unsigned)(real + tail), per = (btot + 7u) / 8u;
                first_surplus = 8 * (int64_t)per;
                if (blk_ < first_surplus) {
                    const unsigned v = ((unsigned)blk_ & 7u) * per + ((unsigned)blk_ >> 3);
                    if (v >= btot) return;
                    const unsigned head = (unsigned)rb * all;
                    unsigned pt, l;
                    if (v < head) { pt = v / all; l = v - pt * all; }
                    else { const unsigned t = v - head, q = t / vg; pt = (unsigned)rb + q; l = (unsigned)groups + (t - q * vg); }
                    blk_ = (int64_t)l * bpg + pt;
                    mapped = true;
                }
            } else if (blk_ < real) {
                blk_ = (blk_ / rb) * bpg + blk_ % rb;
                mapped = true;
            } else if (blk_ < real + tail) {
                blk_ = groups * bpg + (blk_ - real);
                mapped = true;
            }
            if (!mapped) {
                const int64_t s_ = blk_ - first_surplus, db = bpg - rb;  // all-padding block s_ % db of group s_ / db
                if (db > 0 && s_ < groups * db) {
                    const int64_t r0 = ((s_ / db) * bpg + rb + s_ % db) * (kWaves * 16);
                    const int n_out = g.d.n_out;
                    for (int i = tid; i < kWaves * 16 * n_out; i += kWaves * 64) g.out[r0 * n_out + i] = 0.0f;
                }
                return;  // (uniform, before any barrier or LDS-DMA request)
            }
        } else if (g.pm_period > 0) {
            // point-major order of a row set whose groups need not be multiples of 64 rows: group l owns the blocks that START in
            // it, fb(l) = ceil(l P / 64) .. fb(l + 1) - 1 -- nfull = P / 64 of them, plus one more for ceil(G r / 64) of the groups
            // (r = P % 64; the i-th such group is floor(64 i / r)).  Order: tile 0 of every group, tile 1, ..., then the extra ones.
            const unsigned btot = (unsigned)((g.n_rows + kWaves * 16 - 1) / (kWaves * 16)), per = (btot + 7u) / 8u;
            const unsigned v = ((unsigned)blk_ & 7u) * per + ((unsigned)blk_ >> 3);
            if (v >= btot) return;
            const unsigned G = (unsigned)g.pm_groups, nfull = (unsigned)(g.pm_period / (kWaves * 16)), r = (unsigned)(g.pm_period % (kWaves * 16));
            unsigned pt, l;
            if (v < nfull * G) { pt = v / G; l = v - pt * G; }
            else { pt = nfull; l = (unsigned)(((uint64_t)(v - nfull * G) * (kWaves * 16)) / r); }
            blk_ = (int64_t)l * nfull + ((int64_t)l * r + kWaves * 16 - 1) / (kWaves * 16) + pt;
        }
    }
    int64_t row_ = blk_ * (kWaves * 16) + wave * 16 + lj;
    int64_t m_ray = 0;  // SRC == 3: this workgroup's ray and the step of this lane's row
    int m_step = 0;
    if constexpr (SRC == 3) {
        const int64_t n_rays = g.n_rows / g.n_steps;
        m_ray = (int64_t)blockIdx.x % n_rays;
        const int blk = (int)((int64_t)blockIdx.x / n_rays);
        // (agent scope: the flag was raised by a workgroup that may have run on another XCD, whose L2 is not coherent with ours)
        // The flag holds INT_MAX - (lowest block index that saw the ray's first sign change), 0 = none yet: only blocks BEHIND
        // that block may leave.  (A plain boolean would let block b leave when block b + 1 -- dispatched earlier on another
        // XCD -- raised it first, and first_crossing would then read block b's unwritten values: HIP orders nothing between
        // workgroups.)
        if (blk > 0 && g.skip != nullptr) {
            const int s = __hip_atomic_load(g.skip + m_ray, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s != 0 && 0x7fffffff - s < blk) return;
        }
        m_step = blk * (kWaves * 16) + wave * 16 + lj;
        row_ = m_ray * g.n_steps + m_step;
        if (g.n_blocks != nullptr && threadIdx.x == 0) atomicAdd(g.n_blocks, 1ull);
    }
    const int64_t row = row_;
    int64_t n_rows_eff = g.n_rows;
    if constexpr (SRC == 2) {
        // indirect launch (psn_mlp_infer_pe_indirect): the grid covers the capacity of a compacted point list whose length only
        // the device knows; workgroups behind it leave here (uniformly, before any barrier)
        if (g.n_rows_dev != nullptr) {
            n_rows_eff = (int64_t)*g.n_rows_dev;
            if ((int64_t)blockIdx.x * (kWaves * 16) >= n_rows_eff) return;
        }
    }
    const int64_t rowc = row < n_rows_eff ? row : n_rows_eff - 1;
    const int n_layers = g.d.n_layers;

    {  // prefetch the first weight stage (layer 0 may be evaluated entirely through the init tables)
        const int l0 = (g.d.layers[0].n_kt_in + g.d.layers[0].n_kt_act > 0) ? 0 : 1;
        const PsnMlpLayer& L0 = g.d.layers[l0];
        stage_load<2 * NMT, kAsmDma>(g.w + L0.w_off, smem, wave, lane);
    }

    // ---- input features -> registers (MFMA B-operand layout): 16-feature tile t, register r = feature 16t+4g+r
    const int64_t ia = (rowc / g.a_div) % g.a_mod;
    const int64_t ib = (rowc / g.b_div) % g.b_mod;
    const float* init_a_row = g.init_a != nullptr ? g.init_a + ia * (int64_t)g.d.init_stride : nullptr;
    const float* init_b_row = g.init_b != nullptr ? g.init_b + ib * (int64_t)g.d.init_stride : nullptr;
    const float* pa = g.ta != nullptr ? g.ta + ia * (int64_t)(g.d.in_kt_a * 32) : nullptr;
    const float* pb = (g.d.in_kt_b > 0 && g.tb != nullptr) ? g.tb + ib * (int64_t)(g.d.in_kt_b * 32) : nullptr;
    // The (L2-resident) input features are re-read at each layer that consumes them: keeping them in registers for
    // the whole kernel would cost the 32 VGPRs that the second weight-fragment set of stage_compute needs.
    auto load_xin = [&](floatx4 (&xin)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float* src = nullptr;
            if (t < 2 * g.d.in_kt_a) src = pa != nullptr ? pa + t * 16 : nullptr;
            else if (t < 2 * (g.d.in_kt_a + g.d.in_kt_b)) src = pb != nullptr ? pb + (t - 2 * g.d.in_kt_a) * 16 : nullptr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src != nullptr) v = *reinterpret_cast<const float4*>(src + 4 * lg);
            xin[t][0] = v.x;
            xin[t][1] = v.y;
            xin[t][2] = v.z;
            xin[t][3] = v.w;
        }
    };

    // SRC == 1: the positional encoding of the current query point, straight into the B-operand registers; the same
    // expressions as pe_encode_kernel (csrc/pe.hip), so the values are bit-identical to the table path
    float qx = 0.f, qy = 0.f, qz = 0.f;  // current query point of this lane's ray
    // (one 16-column tile t at a time, right before the stage that consumes it: the whole 64-column block held at once
    //  costs 32 registers next to the 64 accumulators and the weight fragments, and spilled)
    auto compute_xin_tile = [&](int t) {
        const int width = 3 + 6 * g.pe_octaves;
        floatx4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
            const int col = 16 * t + 4 * lg + r;
            if (col < 3) {
                v = (col == 0 ? qx : (col == 1 ? qy : qz)) * g.pe_scale;
            } else if (col < width) {
                const int q = col - 3;
                const int f = q / 6, w = q - 6 * f;
                const int c = w % 3;
                const float arg = ldexpf((c == 0 ? qx : (c == 1 ? qy : qz)) * g.pe_scale, f);
                // one range reduction for both (the lanes of a wave want different ones, so `w >= 3 ? cosf : sinf` evaluates
                // both in full); sincosf returns the bits of sinf and cosf (tests: equality with the pe_encode table path)
                float sn, cs;
                sincosf(arg, &sn, &cs);
                v = (w >= 3) ? cs : sn;
            }
            x[r] = v;
        }
        return x;
    };

    for (int i = tid; i < g.n_bias; i += kWaves * 64) bias_lds[i] = g.b[i];  // visible after the first stage barrier
    // The B-side init row (stage 2: T[l] = W_b pe(l) + b of the workgroup's light) is the SAME for all 64 rows of a workgroup that
    // does not straddle two B rows: it then goes through LDS (behind the biases, when there is room) instead of 16 global loads per
    // lane and init layer -- a VMEM instruction in an MFMA wave costs about as much as an MFMA whether it hits L2 or not.
    const float* tb_lds = nullptr;
    if constexpr (!CHAIN && SRC == 0) {
        if (g.init_b != nullptr && g.tb_lds != 0 && g.n_bias + g.d.init_stride <= PSN_MLP_MAX_LAYERS * 256) {
            const int64_t r_first = blk_ * (kWaves * 16), r_last_ = r_first + kWaves * 16 - 1;
            const int64_t r_last = r_last_ < n_rows_eff ? r_last_ : n_rows_eff - 1;
            const int64_t ib0 = (r_first / g.b_div) % g.b_mod;
            if (ib0 == (r_last / g.b_div) % g.b_mod) {  // (uniform over the workgroup)
                float* dst = bias_lds + PSN_MLP_MAX_LAYERS * 256 - g.d.init_stride;
                const float* src = g.init_b + ib0 * (int64_t)g.d.init_stride;
                for (int i = tid; i < g.d.init_stride; i += kWaves * 64) dst[i] = src[i];
                tb_lds = dst;
            }
        }
    }

    floatx4 act[NMT];
    floatx4 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[mt][r] = 0.f;
    if constexpr (CHAIN) {
        if (g.act_init != nullptr) {
            const bool has_row = rowc < g.act_init_rows;
            const float* ap = g.act_init + (has_row ? rowc : 0) * W + 4 * lg;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                float4 t = *reinterpret_cast<const float4*>(ap + mt * 16);
                act[mt][0] = has_row ? t.x : 0.f; act[mt][1] = has_row ? t.y : 0.f; act[mt][2] = has_row ? t.z : 0.f; act[mt][3] = has_row ? t.w : 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) asm volatile("" : "+v"(act[mt]));  // arrived before the first stage's LDS-DMA pieces (see load_xin)
        }
    }

    // Hidden layers (16 output tiles) run in the loop; the final layer (2 output tiles) is peeled off below so
    // that the accumulators never meet a control-flow merge between differently shaped code paths (which
    // makes hipcc shuttle every accumulator register between VGPRs and AGPRs per stage).
    int gstage = 0;  // global stage counter -> LDS buffer parity
    int pending_dump = 0;  // activation-dump stores (0 / 16 / 32) issued after the last LDS-DMA batch
    int pending_stages = 0;  // stage starts (0 or 1) that may leave `pending_dump` stores in flight
    const bool dump_row = row < n_rows_eff && row >= g.save_row0;
    // A wave none of whose rows is dumped skips the store instructions altogether (s_cbranch_execz), so it must not
    // leave room for them in its counted waits -- it would then not wait for its LDS-DMA pieces either.
    const bool wave_dumps = __builtin_amdgcn_ballot_w64(dump_row) != 0;

#define PSN_STAGE(NMT, B0, B1, S_IDX)                                                                       \
    {                                                                                                       \
        /* this wave's LDS-DMA pieces have landed; activation dumps issued after them may stay in flight */ \
        wait_for_weights<NMT>(pending_stages > 0 ? pending_dump : 0);                                       \
        if (pending_stages > 0) --pending_stages;                                                           \
        /* every wave's pieces landed; the other buffer is no longer being read (its fragments fed MFMAs that have  \
           issued).  CHAIN: a bare s_barrier -- __syncthreads() is fence + barrier, and its workgroup-scope release \
           is lowered to s_waitcnt vmcnt(0): every stage barrier then waited for ALL of the wave's outstanding     \
           dump stores, i.e. the counted wait above was void and each layer's dumps were serialised with the next  \
           layer's first stage */                                                                                  \
        if constexpr (CHAIN) asm volatile("s_barrier" ::: "memory");                                         \
        else __syncthreads();                                                                               \
        const int s_ = (S_IDX);                                                                             \
        float* nxt = smem + ((gstage + 1) & 1) * kStageFloats;                                              \
        const float4* wl = reinterpret_cast<const float4*>(smem + (gstage & 1) * kStageFloats);             \
        /* branch-free request (a branch would end the scheduling region the pieces are spread over): after the very \
           last stage the idle buffer receives a copy of the first one */                                   \
        const float* src_ = s_ + 1 < n_st ? wl_g + (int64_t)(s_ + 1) * stage_floats : (next_w != nullptr ? next_w : g.w); \
        if constexpr (X3) stage_compute_x3<NMT, NMT, CHAIN>(acc, B0, B1, wl, lane, [&](int j_) { stage_piece<2 * (NMT), kAsmDma>(src_, nxt, wave, lane, j_); }); \
        else stage_compute<NMT, NMT, kAsmDma>(acc, B0, B1, wl, lane, [&](int j_) { stage_piece<2 * (NMT), kAsmDma>(src_, nxt, wave, lane, j_); }); \
        ++gstage;                                                                                           \
    }

    const int n_hidden = g.d.n_out > 0 ? n_layers - 1 : n_layers;  // n_out == 0: no final layer (backward chains)
    // SRC == 1: bracket and ray of this lane's row (replicated over the four lane groups of a row)
    float ox = 0.f, oy = 0.f, oz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, dl = 0.f, dh = 0.f, fl = 0.f, fh = 0.f, dp = 0.f;
    if constexpr (SRC == 1) {
        ox = g.ray_o[rowc * 3 + 0]; oy = g.ray_o[rowc * 3 + 1]; oz = g.ray_o[rowc * 3 + 2];
        vx = g.ray_d[rowc * 3 + 0]; vy = g.ray_d[rowc * 3 + 1]; vz = g.ray_d[rowc * 3 + 2];
        dl = g.bracket[rowc]; dh = g.bracket[g.n_rows + rowc]; fl = g.bracket[2 * g.n_rows + rowc]; fh = g.bracket[3 * g.n_rows + rowc];
        dp = (-fl) * (dh - dl) / (fh - fl) + dl;  // rendering.py:526 (same expression as secant_step_kernel)
    }
    // SRC == 2 (psn_mlp_infer_pe): the rows are query points, encoded ONCE here -- before accumulators or weight fragments are
    // live, so the branchy sinf / cosf code has the whole register file -- and the 64 columns stay in 16 registers for the
    // layers that consume them (layer 0 and the skip layer), which the 232-register lean loop has room for.
    floatx4 xq[4];
    if constexpr (SRC == 2 || SRC == 3) {
        if constexpr (SRC == 3) {
            // sample_points_kernel (csrc/sample.hip), miss profile: d = near (1 - u) + far u, p = origin + dir d; products and
            // sums rounded separately (-ffp-contract=off), so the point has the bits of the table the two-launch path wrote
            const float d = g.near * g.omu[m_step] + g.far[m_ray] * g.u[m_step];
            qx = g.ray_o[m_ray * 3 + 0] + g.ray_d[m_ray * 3 + 0] * d;
            qy = g.ray_o[m_ray * 3 + 1] + g.ray_d[m_ray * 3 + 1] * d;
            qz = g.ray_o[m_ray * 3 + 2] + g.ray_d[m_ray * 3 + 2] * d;
        } else {
            qx = g.ray_o[rowc * 3 + 0]; qy = g.ray_o[rowc * 3 + 1]; qz = g.ray_o[rowc * 3 + 2];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) xq[t] = compute_xin_tile(t);
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(xq[t]));  // (and the loads behind them) done before the first LDS-DMA pieces
    }
    const int n_iter = SRC == 1 ? g.n_iter : 1;
    for (int iter = 0; iter < n_iter; ++iter) {
    if constexpr (SRC == 1) {  // rendering.py:531: p_mid = ray0 + d_pred * direction
        qx = ox + dp * vx; qy = oy + dp * vy; qz = oz + dp * vz;
    }
    int li = 0;
    for (; li < n_hidden; ++li) {
        const PsnMlpLayer L = g.d.layers[li];
        const int n_st = L.n_kt_in + L.n_kt_act;
        const int stage_floats = NMT * 512;
        const float* wl_g = g.w + L.w_off;
        const float* next_w = li + 1 < n_layers ? g.w + g.d.layers[li + 1].w_off : nullptr;
        {  // bias -> accumulator init
            const float* bp = bias_lds + L.b_off;
            if (li == 0) __syncthreads();  // bias_lds written above
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                float4 bv = *reinterpret_cast<const float4*>(bp + mt * 16 + 4 * lg);
                acc[mt][0] = bv.x;
                acc[mt][1] = bv.y;
                acc[mt][2] = bv.z;
                acc[mt][3] = bv.w;
            }
            if (L.init_off >= 0) {  // per-row precomputed partial products of the input-feature block
                if (init_a_row != nullptr) {
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        float4 u = *reinterpret_cast<const float4*>(init_a_row + L.init_off + mt * 16 + 4 * lg);
                        acc[mt][0] += u.x;
                        acc[mt][1] += u.y;
                        acc[mt][2] += u.z;
                        acc[mt][3] += u.w;
                    }
                }
                if (tb_lds != nullptr) {  // (written before the `li == 0` barrier above, like the biases)
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        const float4 u = *reinterpret_cast<const float4*>(tb_lds + L.init_off + mt * 16 + 4 * lg);
                        acc[mt][0] += u.x;
                        acc[mt][1] += u.y;
                        acc[mt][2] += u.z;
                        acc[mt][3] += u.w;
                    }
                } else if (init_b_row != nullptr) {
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        float4 u = *reinterpret_cast<const float4*>(init_b_row + L.init_off + mt * 16 + 4 * lg);
                        acc[mt][0] += u.x;
                        acc[mt][1] += u.y;
                        acc[mt][2] += u.z;
                        acc[mt][3] += u.w;
                    }
                }
                if constexpr (CHAIN) {
                    if (g.rk_coef != nullptr) {
                        for (int c = 0; c < g.rk_k; ++c) {
                            const float cf = g.rk_coef[rowc * g.rk_k + c];
                            const float* bp2 = g.rk_basis + (int64_t)c * g.d.init_stride + L.init_off + 4 * lg;
#pragma unroll
                            for (int mt = 0; mt < NMT; ++mt) {
                                const float4 u = *reinterpret_cast<const float4*>(bp2 + mt * 16);
                                acc[mt][0] = fmaf(cf, u.x, acc[mt][0]);
                                acc[mt][1] = fmaf(cf, u.y, acc[mt][1]);
                                acc[mt][2] = fmaf(cf, u.z, acc[mt][2]);
                                acc[mt][3] = fmaf(cf, u.w, acc[mt][3]);
                            }
                        }
                    }
                }
            }
        }
        // K tiles from the previous activations first, then from the input features (matches the packer): once the
        // activation tiles are consumed their 64 registers are dead, so the (L2-resident) input features are only
        // fetched then -- the two operand sets never compete for registers with the double-buffered weight fragments.
        // (n_kt_act may be smaller than the width: a layer whose input has 217 real columns -- the one behind the 217-output
        //  layer of the stage-1 geometry network -- reads 7 of the 8 activation k-tiles; the eighth multiplies zeros.
        //  Only the LAST k-tile is conditional: a branch around every stage made hipcc generate slower stage code throughout.)
        if (L.n_kt_act > 0) {
#pragma unroll
            for (int kt = 0; kt < NMT / 2 - 1; ++kt) PSN_STAGE(NMT, act[2 * kt], act[2 * kt + 1], kt)
            if (!TRIM || L.n_kt_act == NMT / 2) PSN_STAGE(NMT, act[NMT - 2], act[NMT - 1], NMT / 2 - 1)
        }
        if constexpr (SRC != 0) {
            if (L.n_kt_in > 0) {  // == 2 (host check): the 64-column encoding
                if constexpr (SRC == 2 || SRC == 3) {
                    PSN_STAGE(NMT, xq[0], xq[1], L.n_kt_act)
                    PSN_STAGE(NMT, xq[2], xq[3], L.n_kt_act + 1)
                } else {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const floatx4 x0 = compute_xin_tile(2 * kt), x1 = compute_xin_tile(2 * kt + 1);
                        PSN_STAGE(NMT, x0, x1, L.n_kt_act + kt)
                    }
                }
            }
        } else if (L.n_kt_in > 0) {
            floatx4 xin[8];
            {
                load_xin(xin);
                // the features must have arrived BEFORE the stage issues its LDS-DMA pieces: those are asm statements the
                // compiler's vmcnt bookkeeping does not see, so a wait placed behind them would be a full vmcnt(0)
#pragma unroll
                for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(xin[t]));
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt < L.n_kt_in) PSN_STAGE(NMT, xin[2 * kt], xin[2 * kt + 1], L.n_kt_act + kt)
            }
        }
        // activation: accumulators become the next layer's B operands.  Optional row-major operands (aux1, aux2)
        // are re-read and optional results (the new activation, a second value) are dumped per layer; the stores
        // are issued behind the already-requested next weight stage and complete under the next layer's MFMAs.
        if constexpr (!CHAIN) {
            // the code is wave-uniform: branch once per layer, never per element (a branch-free softplus inside
            // the element loop gets if-converted and then runs for the ReLU networks too)
            if (L.act == PSN_ACT_RELU) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[mt][r] = relu1(acc[mt][r]);
            } else if (L.act == PSN_ACT_SOFTPLUS100) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 a, unused;
                        softplus100_pair<false>(f32x2{acc[mt][2 * h], acc[mt][2 * h + 1]}, a, unused);
                        act[mt][2 * h] = a.x; act[mt][2 * h + 1] = a.y;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) act[mt] = acc[mt];
            }
            if (g.save[li] != nullptr) {
                const bool bits = g.save_bits[li] != nullptr;
                if (dump_row) {
                    float* dst = g.save[li] + (row - g.save_row0) * W + 4 * lg;
                    // (sign bits: the dumped value is relu(z) or a softplus, > 0 exactly when it compares so; two 32-bit halves
                    //  built tile by tile beside the stores -- formed as a block they cost the kernel 23 registers)
                    unsigned w0 = 0u, w1 = 0u;
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        *reinterpret_cast<float4*>(dst + mt * 16) = make_float4(act[mt][0], act[mt][1], act[mt][2], act[mt][3]);
                        if (bits) {
                            unsigned t = (act[mt][0] > 0.0f ? 1u : 0u) | (act[mt][1] > 0.0f ? 2u : 0u) | (act[mt][2] > 0.0f ? 4u : 0u) | (act[mt][3] > 0.0f ? 8u : 0u);
                            if (mt < 8) w0 |= t << (4 * mt); else w1 |= t << (4 * (mt - 8));
                            asm volatile("" : "+v"(w0), "+v"(w1));
                        }
                    }
                    if (bits) *reinterpret_cast<uint2*>(g.save_bits[li] + (row - g.save_row0) * 4 + lg) = make_uint2(w0, w1);
                }
                pending_dump = wave_dumps ? NMT + (bits ? 1 : 0) : 0;
                pending_stages = wave_dumps ? 1 : 0;
            }
        } else
        {
            const float* p1 = g.mask[li] != nullptr ? g.mask[li] + rowc * W + 4 * lg : nullptr;
            const float* p2 = g.aux2[li] != nullptr ? g.aux2[li] + rowc * W + 4 * lg : nullptr;
            float* d1 = (g.save[li] != nullptr && dump_row) ? g.save[li] + (row - g.save_row0) * W + 4 * lg : nullptr;
            float* d2 = (g.save2[li] != nullptr && dump_row) ? g.save2[li] + (row - g.save_row0) * W + 4 * lg : nullptr;
            // one straight-line body per code (operand loads batched up front, results stored tile by tile)
            const uint32_t m1 = g.save_tiles[li], m2 = g.save2_tiles[li];
#define PSN_CASE(C) case C: chain_activation<C, NMT, FROMA>(acc, act, p1, p2, d1, d2, m1, m2); break;
            switch (L.act) {
                case PSN_ACT_RELU_BITS:  // mask[li] = the forward launch's sign-bit words [n_rows, 4] uint64 (save_bits)
                    chain_activation<PSN_ACT_RELU_BITS, NMT>(acc, act, nullptr, nullptr, d1, d2, m1, m2,
                                                             reinterpret_cast<const unsigned long long*>(g.mask[li])[rowc * 4 + lg]);
                    break;
                PSN_CASE(PSN_ACT_RELU)
                PSN_CASE(PSN_ACT_SOFTPLUS100)
                PSN_CASE(PSN_ACT_RELU_MASK)
                PSN_CASE(PSN_ACT_MUL_AUX)
                PSN_CASE(PSN_ACT_MUL2)
                PSN_CASE(PSN_ACT_SOFTPLUS_BWD)
                PSN_CASE(PSN_ACT_HEAD)
                default: chain_activation<PSN_ACT_NONE, NMT>(acc, act, p1, p2, d1, d2, m1, m2); break;
            }
#undef PSN_CASE
            // (a partially masked dump issues fewer stores than the counted wait of the next stage would leave room for:
            //  that wait then drains the queue instead -- at most two layers per launch)
            const uint32_t all_ = (1u << NMT) - 1u;
            const bool partial = (g.save[li] != nullptr && (m1 & all_) != all_) || (g.save2[li] != nullptr && (m2 & all_) != all_);
            pending_dump = (wave_dumps && !partial) ? (g.save[li] != nullptr ? NMT : 0) + (g.save2[li] != nullptr ? NMT : 0) : 0;
            pending_stages = pending_dump > 0 ? 1 : 0;
        }
    }
    bool wide_final = false;
    if constexpr (CHAIN && NMT == 16) wide_final = g.d.n_out > 32;
    if (wide_final) {
        // final layer with 33..64 outputs (the last layer of the stage-1 gradient sweep, W_0^T: 256 -> 39 encoding columns,
        // which as a hidden-type layer multiplied 16 output tiles to keep 3): four 16-wide output tiles, the 64 KB of weights
        // as two 32 KB stages of 4 k-tiles each
        const PsnMlpLayer L = g.d.layers[li];
        const float* bp = bias_lds + L.b_off;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float4 bv = *reinterpret_cast<const float4*>(bp + mt * 16 + 4 * lg);
            acc[mt][0] = bv.x; acc[mt][1] = bv.y; acc[mt][2] = bv.z; acc[mt][3] = bv.w;
        }
        wait_for_weights<NMT>(pending_stages > 0 ? pending_dump : 0);
        asm volatile("s_barrier" ::: "memory");
        const float4* wl = reinterpret_cast<const float4*>(smem + (gstage & 1) * kStageFloats);
        stage_load<2 * NMT, kAsmDma>(g.w + L.w_off + kStageFloats, smem + ((gstage + 1) & 1) * kStageFloats, wave, lane);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if constexpr (X3) stage_compute_x3<4, NMT, false>(acc, act[2 * kt], act[2 * kt + 1], wl + kt * 512, lane, [](int) {});
            else stage_compute<4, NMT, kAsmDma>(acc, act[2 * kt], act[2 * kt + 1], wl + kt * 512, lane, [](int) {});
        }
        ++gstage;
        wait_for_weights<NMT>(0);
        asm volatile("s_barrier" ::: "memory");
        wl = reinterpret_cast<const float4*>(smem + (gstage & 1) * kStageFloats);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if constexpr (X3) stage_compute_x3<4, NMT, false>(acc, act[8 + 2 * kt], act[8 + 2 * kt + 1], wl + kt * 512, lane, [](int) {});
            else stage_compute<4, NMT, kAsmDma>(acc, act[8 + 2 * kt], act[8 + 2 * kt + 1], wl + kt * 512, lane, [](int) {});
        }
    } else if (g.d.n_out > 0) {  // final layer: 32 (padded) outputs = two 16-wide tiles; all 8 k-tiles arrive as ONE 32 KB stage
        const PsnMlpLayer L = g.d.layers[li];
        const float* bp = bias_lds + L.b_off;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4 bv = *reinterpret_cast<const float4*>(bp + mt * 16 + 4 * lg);
            acc[mt][0] = bv.x;
            acc[mt][1] = bv.y;
            acc[mt][2] = bv.z;
            acc[mt][3] = bv.w;
        }
        wait_for_weights<NMT>(pending_stages > 0 ? pending_dump : 0);
        __syncthreads();
        const float4* wl = reinterpret_cast<const float4*>(smem + (gstage & 1) * kStageFloats);
        if constexpr (SRC == 1) {
            // the next iteration starts over with layer 0: request its first weight stage into the buffer that the
            // last hidden stage has just released (every wave is past the barrier above)
            if (iter + 1 < n_iter) stage_load<2 * NMT, kAsmDma>(g.w + g.d.layers[0].w_off, smem + ((gstage + 1) & 1) * kStageFloats, wave, lane);
        }
        if (g.d.n_out <= 16) {
            // <= 16 outputs (the visibility / occupancy logit, RGB): only output tile 0 of the 32-wide final block is needed --
            // half the MFMAs of this layer, the same products in the same order for the outputs that exist
#pragma unroll
            for (int kt = 0; kt < NMT / 2; ++kt) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float4 a = wl[kt * 256 + e * 128 + lane];
                    const floatx4& bs = e ? act[2 * kt + 1] : act[2 * kt];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bs[0], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bs[1], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bs[2], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bs[3], acc[0], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NMT / 2; ++kt) stage_compute<2, NMT, kAsmDma>(acc, act[2 * kt], act[2 * kt + 1], wl + kt * 256, lane, [](int) {});
        }
        if constexpr (SRC == 1) ++gstage;
    }
    if constexpr (SRC == 1) {
        // occupancy of row lj sits in the lane group lg == 0 (output feature 0): rendering.py:537-548
        float fm = sigmoidf_(acc[0][0] * -10.0f) - g.tau;
        fm = __shfl(fm, lj);
        if (fm < 0.0f) { dl = dp; fl = fm; } else { dh = dp; fh = fm; }
        dp = (-fl) * (dh - dl) / (fh - fl) + dl;
    }
    }  // iterations (SRC == 1; a single pass otherwise)
#undef PSN_STAGE
    if constexpr (SRC == 1) {
        if (row < g.n_rows && lg == 0) g.out[row] = dp;
        return;
    }

    // ---- output: feature f = 16*mt + 4*g + r of the final layer ---------------------------------
    if (row < n_rows_eff && g.d.n_out > 0) {
        const int n_out = g.d.n_out;
        int64_t orow = row;
        if constexpr (SRC == 2) {
            if (g.out_rows != nullptr) orow = g.out_rows[row];
        }
#pragma unroll
        for (int mt = 0; mt < ((CHAIN && NMT == 16) ? 4 : 2); ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mt + 4 * lg + r;
                if (f < n_out) {
                    float v = acc[mt][r];
                    if (g.d.out_act == PSN_OUT_SIGMOID) v = sigmoidf_(v);
                    else if (g.d.out_act == PSN_OUT_OCC) v = sigmoidf_(v * -10.0f);
                    g.out[orow * n_out + f] = v;
                }
            }
        }
    }
    if constexpr (SRC == 3) {
        if (g.skip != nullptr) {
            // val_m = occupancy - tau of this block's 64 steps (first_crossing_kernel's expression), exchanged through LDS (behind
            // the bias block: nothing else lives there); a negative product of neighbours = the ray's first sign change lies in
            // this block or before it, and a ray whose first value is not free is decided as well (rendering.py:466,522)
            float* xch = bias_lds + PSN_MLP_MAX_LAYERS * 256;
            if (lg == 0) xch[wave * 16 + lj] = sigmoidf_(acc[0][0] * -10.0f) - g.tau;
            __syncthreads();
            if (tid < kWaves * 16 - 1) {
                const bool hit = (xch[tid] * xch[tid + 1] < 0.0f) || (tid == 0 && m_step == 0 && !(xch[0] < 0.0f));
                if (hit) __hip_atomic_fetch_max(g.skip + m_ray, 0x7fffffff - m_step / (kWaves * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- fused secant root finder, feature-parallel form ---------------------------------------------------------------------
// The row-parallel engine above needs one serial pass through the network per secant iteration whatever the ray count
// (a wave owns 16 rays x all 256 outputs: 128 MFMAs = 4096 matrix-pipe cycles per weight stage and SIMD), and 4096 rays
// are only 64 workgroups: 1.25 ms per train step on a quarter of the chip.  Here a workgroup owns 16 rays and its four
// waves split the OUTPUT tiles (wave w: tiles 4w .. 4w+3, 32 MFMAs per stage), so 4096 rays are 256 workgroups and a
// pass is four times shorter.  The price is an activation exchange per layer (16 rays x 256 features through LDS, one
// barrier); the weights of a wave are private to it, so they go straight from L2 to registers (8 x 1 KB per stage and
// wave, requested one stage ahead) -- no LDS staging, no stage barriers.  Arithmetic and its order per output element
// are those of mlp_infer_kernel<false, 16, 1> (same packed weights, same MFMA, K order, softplus, encoding): the
// refined depths are bit-identical.
constexpr int kFpLd = 260;  // floats per ray in the exchange buffer (256 + 4: rows 16 B aligned, bank offset 4 per ray)
__global__ __launch_bounds__(256) void root_find_fp_kernel(InferArgs g) {
    __shared__ __attribute__((aligned(16))) float xbuf[2][16 * kFpLd];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lg = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * 16 + lj;
    const int64_t rowc = row < g.n_rows ? row : g.n_rows - 1;
    const int n_layers = g.d.n_layers, n_hidden = n_layers - 1;
    const float4* W4 = reinterpret_cast<const float4*>(g.w);

    // fragment (e, m) of stage s of the layer at float offset w_off: float4 index w_off / 4 + s * 2048 + (e * 16 + 4 wave + m) * 64 + lane
    auto frag_ptr = [&](int64_t w_off, int s) { return W4 + w_off / 4 + (int64_t)s * 2048 + (4 * wave) * 64 + lane; };
    auto load_frags = [&](const float4* p, float4 (&f)[8]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int m = 0; m < 4; ++m) f[e * 4 + m] = p[(e * 16 + m) * 64];
    };
    floatx4 acc[4];
    auto stage_mma = [&](const float4 (&f)[8], const floatx4& b0, const floatx4& b1) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const floatx4& bs = e ? b1 : b0;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[e * 4 + m].x, bs[0], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[e * 4 + m].y, bs[1], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[e * 4 + m].z, bs[2], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[e * 4 + m].w, bs[3], acc[m], 0, 0, 0);
        }
    };

    // ray, bracket and first estimate (rendering.py:526), replicated over the lane groups and waves
    const float ox = g.ray_o[rowc * 3 + 0], oy = g.ray_o[rowc * 3 + 1], oz = g.ray_o[rowc * 3 + 2];
    const float vx = g.ray_d[rowc * 3 + 0], vy = g.ray_d[rowc * 3 + 1], vz = g.ray_d[rowc * 3 + 2];
    float dl = g.bracket[rowc], dh = g.bracket[g.n_rows + rowc], fl = g.bracket[2 * g.n_rows + rowc], fh = g.bracket[3 * g.n_rows + rowc];
    float dp = (-fl) * (dh - dl) / (fh - fl) + dl;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    // positional encoding of the query point in B-operand layout: the expressions of mlp_infer_kernel / pe_encode_kernel
    auto compute_xin = [&](floatx4 (&xin)[4]) {
        const int width = 3 + 6 * g.pe_octaves;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.0f;
                const int col = 16 * t + 4 * lg + r;
                if (col < 3) {
                    v = (col == 0 ? qx : (col == 1 ? qy : qz)) * g.pe_scale;
                } else if (col < width) {
                    const int q = col - 3;
                    const int f = q / 6, w = q - 6 * f;
                    const int c = w % 3;
                    const float arg = ldexpf((c == 0 ? qx : (c == 1 ? qy : qz)) * g.pe_scale, f);
                    v = (w >= 3) ? cosf(arg) : sinf(arg);
                }
                xin[t][r] = v;
            }
        }
    };

    float4 fa[8], fb[8];  // weight fragments of the current / next stage (roles alternate; every layer has an even stage count)
    load_frags(frag_ptr(g.d.layers[0].w_off, 0), fa);
    for (int iter = 0; iter < g.n_iter; ++iter) {
        qx = ox + dp * vx; qy = oy + dp * vy; qz = oz + dp * vz;  // rendering.py:531
        for (int li = 0; li < n_hidden; ++li) {
            const PsnMlpLayer L = g.d.layers[li];
            const int n_st = L.n_kt_in + L.n_kt_act;
            const float4* next_first = frag_ptr(g.d.layers[li + 1 < n_hidden ? li + 1 : 0].w_off, 0);  // (after the last hidden layer: layer 0 of the next iteration)
            {  // bias -> accumulators of this wave's four output tiles
                const float* bp = g.b + L.b_off + 64 * wave + 4 * lg;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 16 * m);
                    acc[m][0] = bv.x; acc[m][1] = bv.y; acc[m][2] = bv.z; acc[m][3] = bv.w;
                }
            }
            const float* xr = &xbuf[(li + 1) & 1][lj * kFpLd + 4 * lg];  // activations of the previous layer (written under parity (li - 1) & 1)
#define FP_STAGE2(B0a, B1a, B0b, B1b, S0)                                                             \
            {                                                                                         \
                load_frags((S0) + 1 < n_st ? frag_ptr(L.w_off, (S0) + 1) : next_first, fb);           \
                stage_mma(fa, B0a, B1a);                                                              \
                load_frags((S0) + 2 < n_st ? frag_ptr(L.w_off, (S0) + 2) : next_first, fa);           \
                stage_mma(fb, B0b, B1b);                                                              \
            }
            if (L.n_kt_act > 0) {
#pragma unroll
                for (int kt = 0; kt < 8; kt += 2) {
                    if (kt + 1 < L.n_kt_act) {
                        const floatx4 a0 = ld4(xr + 32 * kt), a1 = ld4(xr + 32 * kt + 16), a2 = ld4(xr + 32 * kt + 32), a3 = ld4(xr + 32 * kt + 48);
                        FP_STAGE2(a0, a1, a2, a3, kt)
                    } else if (kt < L.n_kt_act) {  // odd count (7 k-tiles behind a 217-output layer): one stage, then the sets swap back
                        const floatx4 a0 = ld4(xr + 32 * kt), a1 = ld4(xr + 32 * kt + 16);
                        load_frags(kt + 1 < n_st ? frag_ptr(L.w_off, kt + 1) : next_first, fb);
                        stage_mma(fa, a0, a1);
#pragma unroll
                        for (int i = 0; i < 8; ++i) fa[i] = fb[i];
                    }
                }
            }
            if (L.n_kt_in > 0) {
                floatx4 xin[4];
                compute_xin(xin);
                FP_STAGE2(xin[0], xin[1], xin[2], xin[3], L.n_kt_act)
            }
#undef FP_STAGE2
            // activation, then this wave's 64 features of its 16 rays into the exchange buffer
            float* xw = &xbuf[li & 1][lj * kFpLd + 64 * wave + 4 * lg];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                floatx4 o;
                if (L.act == PSN_ACT_SOFTPLUS100) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 a, unused;
                        softplus100_pair<false>(f32x2{acc[m][2 * h], acc[m][2 * h + 1]}, a, unused);
                        o[2 * h] = a.x; o[2 * h + 1] = a.y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = L.act == PSN_ACT_RELU ? relu1(acc[m][r]) : acc[m][r];
                }
                st4(xw + 16 * m, o);
            }
            __syncthreads();
        }
        // final layer: output row 0 (the occupancy logit) of the 32-wide final tile pair; every wave evaluates it for
        // its own copy of the bracket -- 64 MFMAs instead of a broadcast and another barrier
        {
            const PsnMlpLayer L = g.d.layers[n_hidden];
            const float* xr = &xbuf[(n_hidden + 1) & 1][lj * kFpLd + 4 * lg];
            const float4* wf = W4 + L.w_off / 4 + lane;  // k-tile kt, half e, tile 0: float4 index kt * 256 + e * 128 + lane
            const float4 bv = *reinterpret_cast<const float4*>(g.b + L.b_off + 4 * lg);
            floatx4 c;
            c[0] = bv.x; c[1] = bv.y; c[2] = bv.z; c[3] = bv.w;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float4 a = wf[kt * 256 + e * 128];
                    const floatx4 b = ld4(xr + 32 * kt + 16 * e);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], c, 0, 0, 0);
                }
            }
            // occupancy of ray lj sits in the lane group lg == 0 (output feature 0): rendering.py:537-548
            float fm = sigmoidf_(c[0] * -10.0f) - g.tau;
            fm = __shfl(fm, lj);
            if (fm < 0.0f) { dl = dp; fl = fm; } else { dh = dp; fh = fm; }
            dp = (-fl) * (dh - dl) / (fh - fl) + dl;
        }
        if (n_hidden & 1) __syncthreads();  // odd depth: layer 0 of the next iteration writes the buffer the final layer has just read
    }
    if (wave == 0 && row < g.n_rows && lg == 0) g.out[row] = dp;
}

// W[rows][cols] (or its transpose in memory), zero-extended to [n_mt*32][k_tiles*32] -> stage order
// [kt32][e(2)][mt16][lane][4]:
//   lane (i = lane & 15, g = lane >> 4), component c  <-  W[16*mt16 + i][32*kt32 + 16*e + 4*g + c]
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ W, int64_t ldw, int rows, int cols,
                                                       int transpose, int n_mt, int k_tiles, float* __restrict__ dst) {
    const int nmt16 = 2 * n_mt;
    const int64_t total = (int64_t)n_mt * k_tiles * 1024;  // floats
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int c = (int)(e & 3);
        int lane = (int)((e >> 2) & 63);
        int64_t blk = e >> 8;
        int per_stage = 2 * nmt16;
        int kt = (int)(blk / per_stage);
        int bi = (int)(blk % per_stage);
        int half = bi / nmt16, mt = bi % nmt16;
        int i = lane & 15, gq = lane >> 4;
        const int r = 16 * mt + i, k = 32 * kt + 16 * half + 4 * gq + c;
        float v = 0.0f;
        if (r < rows && k < cols) v = transpose ? W[(int64_t)k * ldw + r] : W[(int64_t)r * ldw + k];
        dst[e] = v;
    }
}

// All weight blocks of a network in one launch (a train step re-packs ~40 blocks after every optimiser step; one launch
// each was ~0.15 ms of launch-bound work).
struct PackGroupArgs {
    PsnPackItem it[PSN_PACK_MAX_ITEMS];
    int64_t start[PSN_PACK_MAX_ITEMS + 1];  // first 256-float chunk of each item
    int n;
};
__global__ __launch_bounds__(256) void mlp_pack_group_kernel(PackGroupArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int64_t)blockIdx.x >= a.start[i + 1]) ++i;
    const PsnPackItem it = a.it[i];
    const int64_t e = ((int64_t)blockIdx.x - a.start[i]) * 256 + threadIdx.x;  // items are multiples of 1024 floats
    const int nmt16 = 2 * it.n_mt;
    const int c = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const int64_t blk = e >> 8;
    const int per_stage = 2 * nmt16;
    const int kt = (int)(blk / per_stage);
    const int bi = (int)(blk % per_stage);
    const int half = bi / nmt16, mt = bi % nmt16;
    const int r = 16 * mt + (lane & 15);
    auto w_at = [&](int k) {
        float v = 0.0f;
        if (r < it.rows && k < it.cols) v = it.transpose ? it.W[(int64_t)k * it.ldw + r] : it.W[(int64_t)r * it.ldw + k];
        return v;
    };
    if (it.format == PSN_W_BF16X2) {
        // word c of plane `half` of this lane's 16 bytes = k-slots 2 c, 2 c + 1 (stage_compute_x3): slot s <-> k = 32 kt + 16 (s / 4)
        // + 4 g + s % 4; plane 0 = the bf16 heads (round to nearest even), plane 1 = the bf16 of the remainders
        const int k0 = 32 * kt + 16 * (c >> 1) + 4 * (lane >> 4) + 2 * (c & 1);
        const float v0 = w_at(k0), v1 = w_at(k0 + 1);
        int w = chain_cvt2(v0, v1);
        if (half == 1) w = chain_cvt2(v0 - __builtin_bit_cast(float, w << 16), v1 - __builtin_bit_cast(float, w & (int)0xFFFF0000));
        reinterpret_cast<int*>(it.dst)[e] = w;
        return;
    }
    it.dst[e] = w_at(32 * kt + 16 * half + 4 * (lane >> 4) + c);
}

}  // namespace psn

extern "C" int psn_mlp_pack_layers(int n_items, const PsnPackItem* items, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= PSN_PACK_MAX_ITEMS, "mlp_pack_layers: n_items=%d", n_items);
    PackGroupArgs a;
    a.n = n_items;
    a.start[0] = 0;
    for (int i = 0; i < n_items; ++i) {
        const PsnPackItem& it = items[i];
        PSN_CHECK_ARG(it.W && it.dst, "mlp_pack_layers: item %d: null pointer", i);
        PSN_CHECK_ARG(it.format == PSN_W_F32 || it.format == PSN_W_BF16X2, "mlp_pack_layers: item %d: unknown format %d", i, it.format);
        PSN_CHECK_ARG(it.n_mt >= 1 && it.n_mt <= 8 && it.k_tiles >= 1 && it.k_tiles <= 12, "mlp_pack_layers: item %d: n_mt=%d k_tiles=%d", i, it.n_mt, it.k_tiles);
        PSN_CHECK_ARG(it.rows >= 1 && it.rows <= it.n_mt * 32 && it.cols >= 1 && it.cols <= it.k_tiles * 32,
                      "mlp_pack_layers: item %d: %d x %d does not fit %d x %d", i, it.rows, it.cols, it.n_mt * 32, it.k_tiles * 32);
        PSN_CHECK_ARG(it.ldw >= (it.transpose ? it.rows : it.cols), "mlp_pack_layers: item %d: ldw too small", i);
        a.it[i] = it;
        a.start[i + 1] = a.start[i] + (int64_t)it.n_mt * it.k_tiles * 4;  // n_mt * k_tiles * 1024 floats / 256
    }
    hipLaunchKernelGGL(mlp_pack_group_kernel, dim3((unsigned)a.start[n_items]), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_pack_layers");
    return PSN_OK;
}

extern "C" int psn_mlp_pack_layer(const float* W, int64_t ldw, int rows, int cols, int transpose, int n_mt, int k_tiles,
                                  float* dst, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(W && dst, "mlp_pack_layer: null pointer");
    PSN_CHECK_ARG(n_mt >= 1 && n_mt <= 8 && k_tiles >= 1 && k_tiles <= 12, "mlp_pack_layer: n_mt=%d k_tiles=%d", n_mt, k_tiles);
    PSN_CHECK_ARG(rows >= 1 && rows <= n_mt * 32 && cols >= 1 && cols <= k_tiles * 32, "mlp_pack_layer: %d x %d does not fit %d x %d",
                  rows, cols, n_mt * 32, k_tiles * 32);
    PSN_CHECK_ARG(ldw >= (transpose ? rows : cols), "mlp_pack_layer: ldw too small");
    int64_t total = (int64_t)n_mt * k_tiles * 1024;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, ldw, rows, cols, transpose, n_mt, k_tiles, dst);
    PSN_CHECK_LAUNCH("mlp_pack_layer");
    return PSN_OK;
}

// Block order of the lean engine's pair row sets (InferArgs.pm_period): 1 = point-tile-major with the group (light) fastest, an
// eighth of that order per XCD (default), 0 = row order.  The results do not depend on it (a row's arithmetic is its own).
static std::atomic<int> g_point_major{[] { const char* e = getenv("PSN_POINT_MAJOR"); return (e != nullptr && e[0] == '0') ? 0 : 1; }()};
extern "C" int psn_mlp_block_order(int point_major) {
    return g_point_major.exchange(point_major != 0 ? 1 : 0, std::memory_order_relaxed);
}

static int mlp_infer_impl(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                          int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                          const float* init_a, const float* init_b, float* const* save_ptrs, int64_t save_row0,
                          const float* const* mask_ptrs, const float* const* aux2_ptrs, float* const* save2_ptrs,
                          const float* act_init, int64_t act_init_rows, const float* rk_coef, const float* rk_basis, int rk_k,
                          const uint32_t* dump_tiles, int64_t n_rows, float* out, const float* live_count, int64_t live_period,
                          unsigned long long* const* save_bits_ptrs, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && packed_b && (out || desc->n_out == 0), "mlp_infer: null pointer");
    const PsnMlpDesc& d = *desc;
    PSN_CHECK_ARG(d.n_layers >= 1 && d.n_layers <= PSN_MLP_MAX_LAYERS, "mlp_infer: n_layers=%d", d.n_layers);
    PSN_CHECK_ARG(d.in_kt_a >= 0 && d.in_kt_b >= 0 && d.in_kt_a + d.in_kt_b <= 4, "mlp_infer: input tiles %d+%d", d.in_kt_a, d.in_kt_b);
    bool uses_in = false, uses_init = false;
    for (int l = 0; l < d.n_layers; ++l) {
        uses_in = uses_in || d.layers[l].n_kt_in > 0;
        uses_init = uses_init || d.layers[l].init_off >= 0;
    }
    PSN_CHECK_ARG(!uses_in || (tab_a && d.in_kt_a >= 1 && (d.in_kt_b == 0 || tab_b)), "mlp_infer: input table missing");
    PSN_CHECK_ARG((rk_coef == nullptr) == (rk_basis == nullptr) && (rk_coef == nullptr ? rk_k == 0 : (rk_k >= 1 && rk_k <= 4)),
                  "mlp_infer: rank-k init needs coefficients, basis and 1 <= k <= 4 (k=%d)", rk_k);
    PSN_CHECK_ARG(!uses_init || ((init_a || rk_coef) && d.init_stride >= 64 && d.init_stride % 4 == 0), "mlp_infer: init table missing");
    PSN_CHECK_ARG((((uintptr_t)rk_basis) & 15) == 0, "mlp_infer: rk_basis must be 16-byte aligned");
    PSN_CHECK_ARG((((uintptr_t)init_a | (uintptr_t)init_b) & 15) == 0, "mlp_infer: init tables must be 16-byte aligned");
    PSN_CHECK_ARG(d.n_out >= 0 && d.n_out <= 64, "mlp_infer: n_out=%d", d.n_out);
    PSN_CHECK_ARG(a_div >= 1 && a_mod >= 1 && (d.in_kt_b == 0 || (b_div >= 1 && b_mod >= 1)), "mlp_infer: bad index map");
    PSN_CHECK_ARG((((uintptr_t)tab_a | (uintptr_t)tab_b | (uintptr_t)packed_w | (uintptr_t)packed_b) & 15) == 0,
                  "mlp_infer: buffers must be 16-byte aligned");
    // hidden width: 256 (8 output tiles of 32) or 128 (4), the same for every hidden layer of the network
    const int hid = (d.n_out > 0 && d.n_layers == 1) ? 8 : d.layers[0].n_mt;
    PSN_CHECK_ARG(hid == 8 || hid == 4 || hid == 2, "mlp_infer: hidden layers must be 256, 128 or 64 wide (n_mt = 8, 4, 2), got n_mt=%d", hid);
    const int width = hid * 32;
    for (int l = 0; l < d.n_layers; ++l) {
        const PsnMlpLayer& L = d.layers[l];
        const bool last = d.n_out > 0 && l == d.n_layers - 1;
        PSN_CHECK_ARG(L.n_mt == (last ? (d.n_out > 32 ? 2 : 1) : hid), "mlp_infer: layer %d n_mt=%d (hidden layers share one width; final: 1, or 2 for 33..64 outputs)", l, L.n_mt);
        PSN_CHECK_ARG(L.n_kt_in >= 0 && L.n_kt_in <= d.in_kt_a + d.in_kt_b, "mlp_infer: layer %d n_kt_in=%d", l, L.n_kt_in);
        PSN_CHECK_ARG(!last || (L.n_kt_in == 0 && L.n_kt_act == hid), "mlp_infer: the final layer reads the hidden activations only");
        PSN_CHECK_ARG(L.b_off == (int64_t)l * width, "mlp_infer: biases must be packed back to back (one hidden width per layer)");
        PSN_CHECK_ARG(L.n_kt_act == 0 || L.n_kt_act == hid || (L.n_kt_act == hid - 1 && hid > 1), "mlp_infer: layer %d n_kt_act=%d (0, n_mt - 1 or n_mt)", l, L.n_kt_act);
        PSN_CHECK_ARG(l > 0 || L.n_kt_act == 0 || act_init != nullptr, "mlp_infer: layer 0 cannot read activations without act_init");
        PSN_CHECK_ARG(L.n_kt_in + L.n_kt_act >= 1 || (l == 0 && L.init_off >= 0 && d.n_layers > 1), "mlp_infer: layer %d has no input", l);
        PSN_CHECK_ARG(L.init_off < 0 || (!last && L.init_off + width <= d.init_stride && L.init_off % 4 == 0), "mlp_infer: layer %d bad init_off", l);
        PSN_CHECK_ARG((L.w_off % 4) == 0 && (L.b_off % 4) == 0, "mlp_infer: layer %d offsets must be multiples of 4 floats", l);
    }
    if (n_rows <= 0) return PSN_OK;
    InferArgs a = {};
    a.d = d; a.w = packed_w; a.b = packed_b; a.ta = tab_a; a.a_div = a_div; a.a_mod = a_mod;
    a.tb = tab_b; a.b_div = b_div > 0 ? b_div : 1; a.b_mod = b_mod > 0 ? b_mod : 1; a.n_rows = n_rows; a.out = out;
    a.init_a = init_a; a.init_b = init_b;
    a.n_bias = d.n_out > 0 ? (d.n_layers - 1) * width + (d.n_out > 32 ? 64 : 32) : d.n_layers * width;
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) a.save[l] = (save_ptrs != nullptr && l < (d.n_out > 0 ? d.n_layers - 1 : d.n_layers)) ? save_ptrs[l] : nullptr;
    a.save_row0 = save_row0;
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) {
        const bool in_range = l < d.n_layers;
        a.mask[l] = (mask_ptrs != nullptr && in_range) ? mask_ptrs[l] : nullptr;
        a.aux2[l] = (aux2_ptrs != nullptr && in_range) ? aux2_ptrs[l] : nullptr;
        a.save2[l] = (save2_ptrs != nullptr && in_range) ? save2_ptrs[l] : nullptr;
        PSN_CHECK_ARG(((((uintptr_t)a.mask[l]) | ((uintptr_t)a.aux2[l]) | ((uintptr_t)a.save2[l])) & 15) == 0,
                      "mlp_infer: aux / dump tensors must be 16-byte aligned");
        if (in_range) {
            const int act = d.layers[l].act;
            const bool need1 = act == PSN_ACT_RELU_MASK || act == PSN_ACT_MUL_AUX || act == PSN_ACT_MUL2 || act == PSN_ACT_SOFTPLUS_BWD || act == PSN_ACT_RELU_BITS ||
                               act == PSN_ACT_MUL_AUX_A || act == PSN_ACT_MUL2_A || act == PSN_ACT_SOFTPLUS_BWD_A;
            const bool need2 = act == PSN_ACT_MUL2 || act == PSN_ACT_SOFTPLUS_BWD || act == PSN_ACT_MUL2_A || act == PSN_ACT_SOFTPLUS_BWD_A;
            PSN_CHECK_ARG(act >= PSN_ACT_NONE && act <= PSN_ACT_SOFTPLUS_BWD_A, "mlp_infer: layer %d unknown activation %d", l, act);
            PSN_CHECK_ARG(!need1 || a.mask[l] != nullptr, "mlp_infer: layer %d needs aux operand 1", l);
            PSN_CHECK_ARG(!need2 || a.aux2[l] != nullptr, "mlp_infer: layer %d needs aux operand 2", l);
            PSN_CHECK_ARG(act != PSN_ACT_HEAD || a.save[l] != nullptr, "mlp_infer: a HEAD layer needs a dump tensor");
            const bool second = act == PSN_ACT_SOFTPLUS100 || act == PSN_ACT_MUL_AUX || act == PSN_ACT_MUL2 || act == PSN_ACT_MUL_AUX_A || act == PSN_ACT_MUL2_A;
            PSN_CHECK_ARG(a.save2[l] == nullptr || second, "mlp_infer: layer %d (activation %d) has no second value to dump", l, act);
        }
    }
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) PSN_CHECK_ARG((((uintptr_t)a.save[l]) & 15) == 0, "mlp_infer: save buffers must be 16-byte aligned");
    const int rows_per_block = kWaves * 16;
    int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    PSN_CHECK_ARG(blocks < (1ll << 31), "mlp_infer: too many rows");
    a.act_init = act_init;
    a.act_init_rows = act_init_rows;
    a.rk_coef = rk_coef; a.rk_basis = rk_basis; a.rk_k = rk_k;
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) {
        a.save_tiles[l] = dump_tiles ? dump_tiles[l] : 0xFFFFFFFFu;
        a.save2_tiles[l] = dump_tiles ? dump_tiles[PSN_MLP_MAX_LAYERS + l] : 0xFFFFFFFFu;
    }
    PSN_CHECK_ARG((((uintptr_t)act_init) & 15) == 0, "mlp_infer: act_init must be 16-byte aligned");
    PSN_CHECK_ARG(act_init == nullptr || (act_init_rows >= 1 && act_init_rows <= n_rows), "mlp_infer: act_init_rows=%lld of %lld rows", (long long)act_init_rows, (long long)n_rows);
    bool chain = act_init != nullptr || rk_coef != nullptr || dump_tiles != nullptr;
    for (int l = 0; l < d.n_layers; ++l)
        chain = chain || d.layers[l].act > PSN_ACT_SOFTPLUS100 || a.save2[l] != nullptr || a.mask[l] != nullptr || a.aux2[l] != nullptr;
    PSN_CHECK_ARG(d.n_out <= 32 || (chain && hid == 8), "mlp_infer: 33..64 outputs are built for the 256-wide chain engine only (n_out=%d)", d.n_out);
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) {
        a.save_bits[l] = (save_bits_ptrs != nullptr && l < (d.n_out > 0 ? d.n_layers - 1 : d.n_layers)) ? save_bits_ptrs[l] : nullptr;
        PSN_CHECK_ARG(a.save_bits[l] == nullptr || (!chain && a.save[l] != nullptr && (((uintptr_t)a.save_bits[l]) & 7) == 0),
                      "mlp_infer: sign-bit words (layer %d) go with an activation dump of a plain forward launch", l);
    }
    PSN_CHECK_ARG(live_count == nullptr || (!chain && d.n_out >= 1 && live_period >= 64 && live_period % 64 == 0 && save_row0 >= 0 &&
                                            save_row0 % live_period == 0 && save_row0 <= n_rows),
                  "mlp_infer_padded: needs the lean variant, outputs, a period that is a multiple of 64 and save_row0 a multiple of the period (period=%lld save_row0=%lld)",
                  (long long)live_period, (long long)save_row0);
    a.live_count = live_count; a.live_period = live_period > 0 ? live_period : 1;
    {
        static const int tb = [] { const char* e = getenv("PSN_TB_LDS"); return (e != nullptr && e[0] == '0') ? 0 : 1; }();
        a.tb_lds = tb;
        // point-major block order (InferArgs.pm_period): pair row sets row = group * P + point of the lean engine (A/B: PSN_POINT_MAJOR=0)
        const int pm = g_point_major.load(std::memory_order_relaxed);
        const bool pair = !chain && a_div == 1 && (tab_b != nullptr || init_b != nullptr) && b_div == a_mod && b_div >= rows_per_block &&
                          b_mod >= 2 && b_mod < (1 << 20) && n_rows == b_div * b_mod && blocks + 8 < (1ll << 31);
        if (pm && pair && (live_count == nullptr || (live_period == b_div && (n_rows - save_row0) % live_period == 0))) {
            a.pm_period = b_div;
            a.pm_groups = (int)b_mod;
            blocks = live_count != nullptr ? blocks + 7 : (blocks + 7) / 8 * 8;  // (an eighth of the order per XCD: the grid is rounded up)
        }
    }
    const size_t lds_bytes = (2 * kStageFloats + PSN_MLP_MAX_LAYERS * 256) * sizeof(float);
    const dim3 grid((unsigned)blocks), block(kWaves * 64);
    hipStream_t st = (hipStream_t)stream;
    bool trim = false;
    for (int l = 0; l < d.n_layers; ++l) trim = trim || (d.layers[l].n_kt_act > 0 && d.layers[l].n_kt_act < hid);
    PSN_CHECK_ARG(!trim || hid == 8, "mlp_infer: n_kt_act = n_mt - 1 is built for the 256-wide networks only");
    // single-dump programs (PSN_ACT_*_A): the FROMA instantiation of the 256-wide chain kernel runs their base programs on a1 = the
    // softplus output; a launch uses either the _A codes or their base codes, never both
    bool from_a = false, base_mul = false;
    for (int l = 0; l < d.n_layers; ++l) {
        int& act = a.d.layers[l].act;
        if (act == PSN_ACT_MUL_AUX_A || act == PSN_ACT_MUL2_A || act == PSN_ACT_SOFTPLUS_BWD_A) {
            from_a = true;
            act = act == PSN_ACT_MUL_AUX_A ? PSN_ACT_MUL_AUX : act == PSN_ACT_MUL2_A ? PSN_ACT_MUL2 : PSN_ACT_SOFTPLUS_BWD;
        } else if (act == PSN_ACT_MUL_AUX || act == PSN_ACT_MUL2 || act == PSN_ACT_SOFTPLUS_BWD) base_mul = true;
    }
    PSN_CHECK_ARG(!from_a || (hid == 8 && chain && !base_mul), "mlp_infer: the single-dump programs (PSN_ACT_*_A) exist for the 256-wide chain engine and do not mix with their base programs");
    const bool x3 = d.w_format == PSN_W_BF16X2;
    PSN_CHECK_ARG(d.w_format == PSN_W_F32 || x3, "mlp_infer: unknown weight format %d", d.w_format);
    PSN_CHECK_ARG(!x3 || hid == 8, "mlp_infer: split-bf16 weight stages (PSN_W_BF16X2) are built for the 256-wide networks only");
    if (x3 && !chain) {
        if (trim) hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 0, true, false, true>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 0, false, false, true>), grid, block, lds_bytes, st, a);
    } else if (x3) {
        if (from_a) {
            if (trim) hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, true, true, true>), grid, block, lds_bytes, st, a);
            else hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, false, true, true>), grid, block, lds_bytes, st, a);
        } else {
            if (trim) hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, true, false, true>), grid, block, lds_bytes, st, a);
            else hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, false, false, true>), grid, block, lds_bytes, st, a);
        }
    } else if (from_a) {
        if (trim) hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, true, true>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, false, true>), grid, block, lds_bytes, st, a);
    } else if (trim) {
        if (chain) hipLaunchKernelGGL((mlp_infer_kernel<true, 16, 0, true>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 0, true>), grid, block, lds_bytes, st, a);
    } else if (hid == 8) {
        if (chain) hipLaunchKernelGGL((mlp_infer_kernel<true, 16>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<false, 16>), grid, block, lds_bytes, st, a);
    } else if (hid == 4) {
        if (chain) hipLaunchKernelGGL((mlp_infer_kernel<true, 8>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<false, 8>), grid, block, lds_bytes, st, a);
    } else {
        if (chain) hipLaunchKernelGGL((mlp_infer_kernel<true, 4>), grid, block, lds_bytes, st, a);
        else hipLaunchKernelGGL((mlp_infer_kernel<false, 4>), grid, block, lds_bytes, st, a);
    }
    PSN_CHECK_LAUNCH("mlp_infer");
    return PSN_OK;
}

extern "C" int psn_mlp_infer(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                             int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                             const float* init_a, const float* init_b, float* const* save_ptrs, int64_t save_row0,
                             const float* const* mask_ptrs, const float* const* aux2_ptrs, float* const* save2_ptrs,
                             const float* act_init, int64_t act_init_rows, const float* rk_coef, const float* rk_basis, int rk_k,
                             const uint32_t* dump_tiles, int64_t n_rows, float* out, void* stream) {
    return mlp_infer_impl(desc, packed_w, packed_b, tab_a, a_div, a_mod, tab_b, b_div, b_mod, init_a, init_b, save_ptrs, save_row0,
                          mask_ptrs, aux2_ptrs, save2_ptrs, act_init, act_init_rows, rk_coef, rk_basis, rk_k, dump_tiles, n_rows, out,
                          nullptr, 0, nullptr, stream);
}

// psn_mlp_infer, plain forward with activation dumps, that ALSO leaves the sign bits of every dumped activation behind:
// save_bits_ptrs[l] [n_rows - save_row0, 4] uint64 (or NULL per layer), word (row, g) bit 4 mt + r = (feature 16 mt + 4 g + r of
// the row's dumped activation > 0).  A ReLU-backward chain (PSN_ACT_RELU_BITS, the words passed as that layer's aux1) then reads
// 32 bytes per row and layer instead of the activation row; live_count / live_period as for psn_mlp_infer_padded (NULL, 0: none).
extern "C" int psn_mlp_infer_bits(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                                  int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                                  const float* init_a, const float* init_b, float* const* save_ptrs,
                                  unsigned long long* const* save_bits_ptrs, int64_t save_row0, int64_t n_rows, float* out,
                                  const float* live_count, int64_t live_period, void* stream) {
    PSN_CHECK_ARG(save_ptrs != nullptr && save_bits_ptrs != nullptr, "mlp_infer_bits: dump and sign-bit pointer arrays are required");
    return mlp_infer_impl(desc, packed_w, packed_b, tab_a, a_div, a_mod, tab_b, b_div, b_mod, init_a, init_b, save_ptrs, save_row0,
                          nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, n_rows, out, live_count, live_period,
                          save_bits_ptrs, stream);
}

// psn_mlp_infer (lean variant: no chain operands) over a PADDED row set: the rows [0, save_row0) come in groups of
// live_period rows (stage 2: one group per shading light over a surface-pixel list padded to a fixed capacity, so that one
// captured HIP graph serves batches with different surface counts), of which the first live_count[0] -- a float on the
// device, psn_surface_index's count -- are real.  Workgroups all of whose 64 rows are padding write zeros and leave; every
// other row is evaluated as by psn_mlp_infer (bit-identical), the rows from save_row0 on all of them.
extern "C" int psn_mlp_infer_padded(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                                    int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                                    const float* init_a, const float* init_b, float* const* save_ptrs, int64_t save_row0,
                                    int64_t n_rows, float* out, const float* live_count, int64_t live_period, void* stream) {
    PSN_CHECK_ARG(live_count != nullptr, "mlp_infer_padded: live_count is null");
    return mlp_infer_impl(desc, packed_w, packed_b, tab_a, a_div, a_mod, tab_b, b_div, b_mod, init_a, init_b, save_ptrs, save_row0,
                          nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, n_rows, out, live_count, live_period, nullptr, stream);
}

// Lean evaluation of a 256-wide network whose input block is the positional encoding of a 3-vector (the stage-1 occupancy
// queries, stage1/model/network.py:141-150 + 85-101): `points` [n_rows, 3] in, the encoding gamma(scale * p) is formed
// in the kernel prologue in the B-operand registers -- the same expressions as pe_encode_kernel, so the result equals
// psn_pe_encode + psn_mlp_infer bit for bit without the [n_rows, 64] table ever existing in HBM.
static int mlp_infer_pe_impl(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* points,
                             int64_t n_rows, const long long* n_rows_dev, const int64_t* out_rows, int pe_octaves, float pe_scale,
                             float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && packed_b && points && out, "mlp_infer_pe: null pointer");
    const PsnMlpDesc& d = *desc;
    PSN_CHECK_ARG(d.n_layers >= 2 && d.n_layers <= PSN_MLP_MAX_LAYERS && d.n_out >= 1 && d.n_out <= 32, "mlp_infer_pe: n_layers=%d n_out=%d", d.n_layers, d.n_out);
    PSN_CHECK_ARG(d.w_format == PSN_W_F32 || d.w_format == PSN_W_BF16X2, "mlp_infer_pe: unknown weight format %d", d.w_format);
    PSN_CHECK_ARG(d.out_act >= PSN_OUT_NONE && d.out_act <= PSN_OUT_OCC, "mlp_infer_pe: out_act=%d", d.out_act);
    PSN_CHECK_ARG(d.in_kt_a == 2 && d.in_kt_b == 0 && 3 + 6 * pe_octaves <= 64 && pe_octaves >= 0,
                  "mlp_infer_pe: the input block is one 64-column positional encoding (in_kt_a = 2), got in_kt_a=%d octaves=%d", d.in_kt_a, pe_octaves);
    PSN_CHECK_ARG(d.layers[0].n_kt_in == 2 && d.layers[0].n_kt_act == 0, "mlp_infer_pe: layer 0 reads the encoding as k-tiles");
    for (int l = 0; l < d.n_layers; ++l) {
        const PsnMlpLayer& L = d.layers[l];
        const bool last = l == d.n_layers - 1;
        PSN_CHECK_ARG(L.n_mt == (last ? 1 : 8) && L.init_off < 0, "mlp_infer_pe: layer %d: 256-wide hidden layers without init tables only", l);
        PSN_CHECK_ARG(L.act == PSN_ACT_SOFTPLUS100 || L.act == PSN_ACT_RELU || L.act == PSN_ACT_NONE, "mlp_infer_pe: layer %d activation", l);
        PSN_CHECK_ARG(L.b_off == (int64_t)l * 256, "mlp_infer_pe: biases must be packed back to back");
        PSN_CHECK_ARG(L.n_kt_in == 0 || L.n_kt_in == 2, "mlp_infer_pe: layer %d n_kt_in=%d", l, L.n_kt_in);
        PSN_CHECK_ARG(last ? (L.n_kt_act == 8 && L.n_kt_in == 0) : (L.n_kt_act == 0 || L.n_kt_act == 7 || L.n_kt_act == 8), "mlp_infer_pe: layer %d n_kt_act=%d", l, L.n_kt_act);
        PSN_CHECK_ARG(L.n_kt_in + L.n_kt_act >= 1, "mlp_infer_pe: layer %d has no input", l);
    }
    if (n_rows <= 0) return PSN_OK;
    InferArgs a = {};
    a.d = d; a.w = packed_w; a.b = packed_b; a.a_div = 1; a.a_mod = 1; a.b_div = 1; a.b_mod = 1; a.n_rows = n_rows; a.out = out;
    a.n_bias = (d.n_layers - 1) * 256 + 32;
    a.ray_o = points; a.pe_scale = pe_scale; a.pe_octaves = pe_octaves; a.n_rows_dev = n_rows_dev; a.out_rows = out_rows;
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) { a.save_tiles[l] = 0xFFFFFFFFu; a.save2_tiles[l] = 0xFFFFFFFFu; }
    const int64_t blocks = (n_rows + kWaves * 16 - 1) / (kWaves * 16);
    PSN_CHECK_ARG(blocks < (1ll << 31), "mlp_infer_pe: too many rows");
    const size_t lds_bytes = (2 * kStageFloats + PSN_MLP_MAX_LAYERS * 256) * sizeof(float);
    if (d.w_format == PSN_W_BF16X2) hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 2, true, false, true>), dim3((unsigned)blocks), dim3(kWaves * 64), lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 2, true>), dim3((unsigned)blocks), dim3(kWaves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_infer_pe");
    return PSN_OK;
}

extern "C" int psn_mlp_infer_pe(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* points,
                                int64_t n_rows, int pe_octaves, float pe_scale, float* out, void* stream) {
    return mlp_infer_pe_impl(desc, packed_w, packed_b, points, n_rows, nullptr, nullptr, pe_octaves, pe_scale, out, stream);
}

// The same network over a COMPACTED point list whose length only the device knows (psn_shadow_points' counter): the grid
// covers `capacity` rows, workgroups behind *n_rows_dev leave in their prologue, and row r's outputs are written to
// out[out_rows[r]] when out_rows is given -- the shadow-ray path of stage1/model/rendering.py:378-408 without a host
// synchronisation between the box test and the network, and without a separate scatter of the occupancies.
extern "C" int psn_mlp_infer_pe_indirect(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* points,
                                         int64_t capacity, const long long* n_rows_dev, const int64_t* out_rows, int pe_octaves,
                                         float pe_scale, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(n_rows_dev != nullptr, "mlp_infer_pe_indirect: n_rows_dev is required");
    return mlp_infer_pe_impl(desc, packed_w, packed_b, points, capacity, n_rows_dev, out_rows, pe_octaves, pe_scale, out, stream);
}

// Ray-march sweep (stage1/model/rendering.py:447-462): the occupancy of n_steps proposal points per ray, points generated and
// encoded in the kernel (no [N, M, 3] point tensor), with early termination behind a ray's first sign change when `skip`
// (int32 [n_rays], ZEROED by the caller) is given.  occ [n_rays, n_steps]: every value up to and including the pair of the
// first sign change is written -- exactly the values psn_first_crossing reads; values of skipped blocks are left untouched.
extern "C" int psn_march_sweep(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* origin,
                               const float* dir, const float* far, const float* u, const float* omu, float near, int64_t n_rays,
                               int n_steps, float tau, int pe_octaves, float pe_scale, int* skip, unsigned long long* n_blocks, float* occ, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && packed_b && origin && dir && far && u && omu && occ, "march_sweep: null pointer");
    const PsnMlpDesc& d = *desc;
    PSN_CHECK_ARG(d.n_layers >= 2 && d.n_layers <= PSN_MLP_MAX_LAYERS && d.n_out == 1 && d.out_act == PSN_OUT_OCC,
                  "march_sweep: expects an occupancy network (one output, PSN_OUT_OCC)");
    PSN_CHECK_ARG(d.w_format == PSN_W_F32 || d.w_format == PSN_W_BF16X2, "march_sweep: unknown weight format %d", d.w_format);
    PSN_CHECK_ARG(d.in_kt_a == 2 && d.in_kt_b == 0 && 3 + 6 * pe_octaves <= 64 && pe_octaves >= 0,
                  "march_sweep: the input block is one 64-column positional encoding (in_kt_a = 2), got in_kt_a=%d octaves=%d", d.in_kt_a, pe_octaves);
    PSN_CHECK_ARG(d.layers[0].n_kt_in == 2 && d.layers[0].n_kt_act == 0, "march_sweep: layer 0 reads the encoding as k-tiles");
    for (int l = 0; l < d.n_layers; ++l) {
        const PsnMlpLayer& L = d.layers[l];
        const bool last = l == d.n_layers - 1;
        PSN_CHECK_ARG(L.n_mt == (last ? 1 : 8) && L.init_off < 0, "march_sweep: layer %d: 256-wide hidden layers without init tables only", l);
        PSN_CHECK_ARG(L.act == PSN_ACT_SOFTPLUS100 || L.act == PSN_ACT_RELU || L.act == PSN_ACT_NONE, "march_sweep: layer %d activation", l);
        PSN_CHECK_ARG(L.b_off == (int64_t)l * 256, "march_sweep: biases must be packed back to back");
        PSN_CHECK_ARG(L.n_kt_in == 0 || L.n_kt_in == 2, "march_sweep: layer %d n_kt_in=%d", l, L.n_kt_in);
        PSN_CHECK_ARG(last ? (L.n_kt_act == 8 && L.n_kt_in == 0) : (L.n_kt_act == 0 || L.n_kt_act == 7 || L.n_kt_act == 8), "march_sweep: layer %d n_kt_act=%d", l, L.n_kt_act);
        PSN_CHECK_ARG(L.n_kt_in + L.n_kt_act >= 1, "march_sweep: layer %d has no input", l);
    }
    PSN_CHECK_ARG(n_steps >= 64 && n_steps % (kWaves * 16) == 0, "march_sweep: n_steps=%d must be a multiple of 64 (one workgroup = 64 steps of one ray)", n_steps);
    if (n_rays <= 0) return PSN_OK;
    InferArgs a = {};
    a.d = d; a.w = packed_w; a.b = packed_b; a.a_div = 1; a.a_mod = 1; a.b_div = 1; a.b_mod = 1; a.n_rows = n_rays * n_steps; a.out = occ;
    a.n_bias = (d.n_layers - 1) * 256 + 32;
    a.ray_o = origin; a.ray_d = dir; a.far = far; a.u = u; a.omu = omu; a.near = near; a.n_steps = n_steps; a.skip = skip; a.n_blocks = n_blocks; a.tau = tau;
    a.pe_scale = pe_scale; a.pe_octaves = pe_octaves;
    for (int l = 0; l < PSN_MLP_MAX_LAYERS; ++l) { a.save_tiles[l] = 0xFFFFFFFFu; a.save2_tiles[l] = 0xFFFFFFFFu; }
    const int64_t blocks = n_rays * (n_steps / (kWaves * 16));
    PSN_CHECK_ARG(blocks < (1ll << 31), "march_sweep: too many rows");
    const size_t lds_bytes = (2 * kStageFloats + PSN_MLP_MAX_LAYERS * 256 + kWaves * 16) * sizeof(float);
    if (d.w_format == PSN_W_BF16X2) hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 3, true, false, true>), dim3((unsigned)blocks), dim3(kWaves * 64), lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 3, true>), dim3((unsigned)blocks), dim3(kWaves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("march_sweep");
    return PSN_OK;
}

// Fused secant refinement of the ray march (stage1/model/rendering.py:525-555) on the occupancy network packed by
// fused.pack_geo_occupancy (256-wide softplus network whose input block is the positional encoding of the point).
extern "C" int psn_root_find(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* origin,
                             const float* dir, const float* bracket, int64_t n_rays, float tau, int n_iter, int pe_octaves,
                             float pe_scale, float* d_out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && packed_b && origin && dir && bracket && d_out, "root_find: null pointer");
    const PsnMlpDesc& d = *desc;
    PSN_CHECK_ARG(d.n_layers >= 2 && d.n_layers <= PSN_MLP_MAX_LAYERS && d.n_out == 1 && d.out_act == PSN_OUT_OCC,
                  "root_find: expects an occupancy network (one output, PSN_OUT_OCC)");
    PSN_CHECK_ARG(d.w_format == PSN_W_F32, "root_find: fp32 weight stages only");
    PSN_CHECK_ARG(d.in_kt_a == 2 && d.in_kt_b == 0 && 3 + 6 * pe_octaves <= 64 && pe_octaves >= 0,
                  "root_find: the input block is one 64-column positional encoding (in_kt_a = 2), got in_kt_a=%d octaves=%d", d.in_kt_a, pe_octaves);
    PSN_CHECK_ARG(d.layers[0].n_kt_in == 2 && d.layers[0].n_kt_act == 0 && d.layers[0].init_off < 0, "root_find: layer 0 reads the encoding as k-tiles");
    for (int l = 0; l < d.n_layers; ++l) {
        const PsnMlpLayer& L = d.layers[l];
        const bool last = l == d.n_layers - 1;
        PSN_CHECK_ARG(L.n_mt == (last ? 1 : 8) && L.init_off < 0, "root_find: layer %d: 256-wide hidden layers without init tables only", l);
        PSN_CHECK_ARG(L.act == (last ? PSN_ACT_NONE : PSN_ACT_SOFTPLUS100) || L.act == PSN_ACT_RELU || L.act == PSN_ACT_NONE, "root_find: layer %d activation", l);
        PSN_CHECK_ARG(L.b_off == (int64_t)l * 256, "root_find: biases must be packed back to back");
    }
    PSN_CHECK_ARG(n_iter >= 0 && n_iter <= 64, "root_find: n_iter=%d", n_iter);
    if (n_rays <= 0) return PSN_OK;
    InferArgs a = {};
    a.d = d; a.w = packed_w; a.b = packed_b; a.a_div = 1; a.a_mod = 1; a.b_div = 1; a.b_mod = 1; a.n_rows = n_rays; a.out = d_out;
    a.n_bias = (d.n_layers - 1) * 256 + 32;
    a.ray_o = origin; a.ray_d = dir; a.bracket = bracket; a.tau = tau; a.pe_scale = pe_scale; a.n_iter = n_iter; a.pe_octaves = pe_octaves;
    const int64_t blocks = (n_rays + kWaves * 16 - 1) / (kWaves * 16);
    PSN_CHECK_ARG(blocks < (1ll << 31), "root_find: too many rays");
    const size_t lds_bytes = (2 * kStageFloats + PSN_MLP_MAX_LAYERS * 256) * sizeof(float);
    // feature-parallel form whenever the network has the shape it is written for (even stage counts per layer: the
    // 64-column encoding = 2 input k-tiles, 8 activation k-tiles); the row-parallel engine otherwise
    bool fp = d.n_layers >= 2;
    for (int l = 0; l < d.n_layers - 1; ++l) fp = fp && (d.layers[l].n_kt_in == 0 || d.layers[l].n_kt_in == 2) && (d.layers[l].n_kt_act == 0 || d.layers[l].n_kt_act >= 7) && d.layers[l].n_kt_act <= 8;
    fp = fp && d.layers[d.n_layers - 1].n_kt_act == 8 && d.layers[d.n_layers - 1].n_kt_in == 0;
    if (const char* ev = getenv("PSN_ROOT_FIND_ROW_PARALLEL")) fp = fp && ev[0] != '1';  // test hook: both forms must agree bit for bit
    if (fp) {
        const int64_t blocks16 = (n_rays + 15) / 16;
        PSN_CHECK_ARG(blocks16 < (1ll << 31), "root_find: too many rays");
        hipLaunchKernelGGL(root_find_fp_kernel, dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((mlp_infer_kernel<false, 16, 1, true>), dim3((unsigned)blocks), dim3(kWaves * 64), lds_bytes, (hipStream_t)stream, a);
    }
    PSN_CHECK_LAUNCH("root_find");
    return PSN_OK;
}
