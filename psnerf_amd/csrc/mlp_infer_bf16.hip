// bf16 inference engine for the 256-wide ReLU networks of stage 2 -- the evaluation / relighting path only
// (BASELINE config 5: "bf16 MFMA path ... envmap relight eval"; stage2/eval.py:199-218 evaluates visibility_net on
// 512 environment lights x every surface pixel, no gradients).  Training and every parity-gated path stay on the
// exact-fp32 engine of mlp_infer.hip; this kernel is opt-in (PSNetwork.inference_precision = 'bf16'; conf train.vis_bf16
// for the detached shading rows of a training step).
//
// Same idea as the fp32 engine -- activations never leave registers, the D registers of one layer are the B
// operands of the next through a permuted K order -- re-tiled for v_mfma_f32_32x32x16_bf16 (16x the fp32 rate):
//   OUT^T[features, rows] = W[features, K] * ACT^T[K, rows],   M = 32 features, N = 32 rows, K = 16 per MFMA.
// A wave owns 64 rows (two N tiles) x 256 features = 256 fp32 accumulators; lane (n = lane & 31, h = lane >> 5)
// holds, for output tile ot and register v = 4q + r, feature 32 ot + 8 q + 4 h + r of row n (MFMA C/D layout).
// After ReLU the pairs (v, v+1) are rounded to bf16 (v_cvt_pk_bf16_f32, RNE; ReLU = v_pk_max_i16 with 0 on the
// packed halves) and registers [8 qp, 8 qp + 8) of tile ot become B operand k-step 2 ot + qp of the next layer, whose
// lane supplies K indices 8 h + j  <->  feature 32 ot + 16 qp + 8 (j / 4) + 4 h + (j % 4): psn_mlp_pack_bf16
// orders the weight columns to match.  The input block [PE(x) | PE(l)] (2 x 64 bf16 per row, gathered from two
// tables) is consumed as 8 natural-order k-steps; the bias rides along as one more k-step whose B operand is the
// constant (1, 1, 0, ...) against the columns (bf16(b), bf16(b - bf16(b))): no VALU add, ~16 significant bits.
//
// Workgroup = 4 waves = 256 rows, one workgroup per CU (1 wave per SIMD, 512 registers): at 16x the MFMA rate the
// weight stream, not the matrix pipe, is what has to be amortised -- 256 rows per pass keep it at ~3.8 KB per row
// from L2 and the fragment reads at a quarter of the LDS bandwidth.  Weights stream through LDS by LDS-DMA in stages
// of 8-9 k-steps (64 / 72 KB), double buffered.  With one wave per SIMD nothing hides a block of non-MFMA work, but
// ~5 single-issue instructions fit into the 32-cycle gap behind each MFMA: stages run output-tile-pair-major and the
// epilogue of a pair is issued in the gaps of the pairs that follow it (bf_stage_mma).
//
// Roofline: MFMA-bound in bf16 (2.5 PFLOP/s dense).  Measured: 1.35 PFLOP/s algorithmic on the visibility net of
// bear.conf (0.54 of the peak), matrix pipe 67 % busy at 2.25 GHz.
#include "common.h"

namespace psn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef short shortx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));

struct Bf16Args {
    PsnBf16Desc d;
    const unsigned char* w;
    const float* final_bias;
    const unsigned char* ta;
    const unsigned char* tb;
    unsigned a_div, a_mod, b_div, b_mod;
    unsigned n_rows;
    float* out;
    // GROUPED: rows (g, n) -> g * rows_per_group + n read table A row n; group g supplies the bias k-steps of the layers
    // that read the input block (gbias [n_groups][n_in_layers][8 KB], psn_bf16_pack_group_bias)
    const unsigned char* gbias;
    unsigned rows_per_group, tiles_per_group, n_in_layers;
};

constexpr int kKsBytes = 8192;               // one k-step of a hidden layer: 8 output tiles x 64 lanes x 16 B
constexpr int kStageBytes = 9 * kKsBytes;    // largest stage: 8 k-steps + the bias k-step
constexpr int kBfWaves = 4;

// LDS-DMA 4 x NPW KB of the weight stream: wave w moves the contiguous blocks [w NPW, (w + 1) NPW) of 1 KB each.  Four
// consecutive blocks share one base (global address and M0) and differ in the instruction offset only, so a stage costs
// NPW global_load_lds plus ~NPW/4 address / M0 updates, no branches: the scheduler can spread them between MFMAs.
// Piece j < NPS: wave w moves the contiguous 1 KB blocks [w NPS, (w + 1) NPS) of the stream part of a stage; pieces
// NPS, NPS + 1: the wave's 2 KB of the stage's bias k-step (source bsrc, LDS slot bias_off), which comes from the weight
// stream or, for a layer whose bias depends on the row group, from that group's table.
template <int NPS>
__device__ __forceinline__ void bf_dma_piece(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, const unsigned char* __restrict__ bsrc,
                                             int bias_off, int wave, int lane, int j) {
    const unsigned voff = lane * 16;
    if (j < NPS) {
        const int grp = j >> 2;
        const unsigned char* base = gsrc + wave * (NPS * 1024) + grp * 4096;      // wave-uniform: SGPR pair
        const unsigned lds = lds_addr(lds_dst + wave * (NPS * 1024) + grp * 4096);  // wave-uniform: M0
        switch (j & 3) {
            case 0: lds_dma_16<0>(base, lds, voff); break;
            case 1: lds_dma_16<1024>(base, lds, voff); break;
            case 2: lds_dma_16<2048>(base, lds, voff); break;
            default: lds_dma_16<3072>(base, lds, voff); break;
        }
    } else {
        const unsigned char* base = bsrc + wave * 2048;
        const unsigned lds = lds_addr(lds_dst + bias_off + wave * 2048);
        if (j == NPS) lds_dma_16<0>(base, lds, voff);
        else lds_dma_16<1024>(base, lds, voff);
    }
}

// ReLU + round-to-nearest-even bf16 of accumulator registers [8 qp + 2 i, 8 qp + 2 i + 2): dword i of the next layer's B operand.
__device__ __forceinline__ int bf_pack_relu2(const floatx16& c, int qp, int i) {
    floatx2 f;
    f[0] = c[8 * qp + 2 * i];
    f[1] = c[8 * qp + 2 * i + 1];
    shortx2 s = __builtin_bit_cast(shortx2, __builtin_convertvector(f, bf16x2));
    const shortx2 z = {0, 0};
    s = __builtin_elementwise_max(s, z);  // negative floats are negative int16: max with 0 is ReLU (and -0 -> +0)
    return __builtin_bit_cast(int, s);
}

// Epilogue of output-tile pair q (tiles 2q, 2q+1, both row tiles) = 8 jobs of 16 VALU instructions (8 accumulator
// reads, 4 cvt, 4 max); job j writes B operand k-step 2 ot + qp of row tile t in place, one dword per part i.
__device__ __forceinline__ void bf_epilogue_part(const floatx16 (&acc)[2][8], bf16x8 (&bact)[2][16], int q, int j, int i) {
    const int ot = 2 * q + (j >> 2), t = (j >> 1) & 1, qp = j & 1;
    intx4 o = __builtin_bit_cast(intx4, bact[t][2 * ot + qp]);
    o[i] = bf_pack_relu2(acc[t][ot], qp, i);
    bact[t][2 * ot + qp] = __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ void bf_epilogue_job(const floatx16 (&acc)[2][8], bf16x8 (&bact)[2][16], int q, int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bf_epilogue_part(acc, bact, q, j, i);
}

enum { BF_SRC_ACT_LO = 0, BF_SRC_ACT_HI = 1, BF_SRC_IN = 2, BF_SRC_INB = 3 };  // B operands of a stage (+ bias k-step: ACT_LO, INB)
enum { BF_EPI_NONE = 0, BF_EPI_FIRST = 1, BF_EPI_LAST = 2 };

// One stage (8 or 9 k-steps) in OUTPUT-TILE-PAIR-major order: pair p = tiles 2p, 2p+1 runs through all k-steps of the
// stage (4 MFMAs per k-step: 2 tiles x 2 row tiles, so consecutive MFMAs on one accumulator are 4 apart) before
// pair p+1 starts.  In the LAST stage of a layer pair p is therefore final after (p+1)/4 of the stage, and its
// epilogue (accumulator -> ReLU -> bf16 B operand, VALU only) is issued in the MFMA gaps of later pairs -- with one
// wave per SIMD nothing else could hide it:
//   LAST  stage of layer l  :  P0 | P1 + E(l,0) | P2 + E(l,1) | P3
//   FIRST stage of layer l+1:  P0 + E(l,2) | P1 + E(l,3) | P2 | P3
// E(l,0..1) produce k-steps 0..7 (all the FIRST stage reads), E(l,2..3) k-steps 8..15 (read by the second stage), and
// they read the accumulators of pairs 2, 3 before this stage's P2 / P3 overwrite them.  All writes are in place:
// a stage never reads the k-steps its jobs write.
// Weight fragments: 2 ds_read_b128 per k-step, ring of 3 (requested two k-steps = 256 MFMA cycles ahead).
template <int SRC, int EPI, int NIN, int NPT, typename RequestPiece>
__device__ __forceinline__ void bf_stage_mma(floatx16 (&acc)[2][8], bf16x8 (&bact)[2][16], const bf16x8 (&bin)[2][8],
                                             const bf16x8& bias_b, const bf16x8* __restrict__ wl, int lane,
                                             RequestPiece request_piece) {
    // k-steps of the stage: 8 activation k-steps (+ bias), or the NIN k-steps of the input block (+ bias for layer 0)
    constexpr int NKS = SRC == BF_SRC_ACT_LO ? 9 : SRC == BF_SRC_ACT_HI ? 8 : SRC == BF_SRC_IN ? NIN : NIN + 1;
    constexpr bool ZERO_C = SRC == BF_SRC_ACT_LO || SRC == BF_SRC_INB;  // first stage of its layer
    constexpr int NSTEP = 4 * NKS;
    constexpr int DMA_PAIR = EPI == BF_EPI_FIRST ? 2 : 0;  // the pair whose gaps carry the LDS-DMA requests (one without jobs)
    constexpr int JPK = NKS >= 8 ? 1 : 2;                  // epilogue jobs per k-step: the 8 jobs of a pair fit into its k-steps
    static_assert(NPT <= 4 * NKS && 8 <= JPK * NKS, "stage too short for its requests / jobs");
    auto bop = [&](int t, int ks) -> bf16x8 {
        if constexpr (SRC == BF_SRC_ACT_LO) return ks < 8 ? bact[t][ks] : bias_b;
        else if constexpr (SRC == BF_SRC_ACT_HI) return bact[t][8 + ks];
        else if constexpr (SRC == BF_SRC_IN) return bin[t][ks];
        else return ks < NIN ? bin[t][ks] : bias_b;
    };
    auto frag = [&](int step, int o) -> bf16x8 { return wl[((step % NKS) * 8 + 2 * (step / NKS) + o) * 64 + lane]; };
    constexpr int R = 3;  // fragment ring: requested R - 1 k-steps = (R - 1) x 128 MFMA cycles ahead (deeper rings measured +-0)
    bf16x8 a[R][2];
#pragma unroll
    for (int i = 0; i < R - 1; ++i) { a[i][0] = frag(i, 0); a[i][1] = frag(i, 1); }
    __builtin_amdgcn_sched_barrier(0);
    floatx16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
        const int p = step / NKS, ks = step % NKS;
        if (step + R - 1 < NSTEP) {
            a[(step + R - 1) % R][0] = frag(step + R - 1, 0);
            a[(step + R - 1) % R][1] = frag(step + R - 1, 1);
        }
        const bf16x8 b0 = bop(0, ks), b1 = bop(1, ks);
        int q = -1;
        if (EPI == BF_EPI_FIRST && p < 2) q = 2 + p;
        if (EPI == BF_EPI_LAST && (p == 1 || p == 2)) q = p - 1;
        // The order of a k-step is pinned by one scheduling region per MFMA (the LDS-DMA pieces are asm statements, and the
        // accumulator reads of a job are COPYs until register allocation: no scheduling group matches either):
        //   [2 fragment reads (two k-steps ahead) | MFMA | piece | job part(s)] [MFMA | piece | part(s)] x 3
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int o = m >> 1, t = m & 1, ot = 2 * p + o;
            acc[t][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step % R][o], t ? b1 : b0, (ZERO_C && ks == 0) ? zero : acc[t][ot], 0, 0, 0);
            if (p == DMA_PAIR && 4 * ks + m < NPT) request_piece(4 * ks + m);  // one request per MFMA until all NPT are out
#pragma unroll
            for (int jj = 0; jj < JPK; ++jj)
                if (q >= 0 && ks * JPK + jj < 8) bf_epilogue_part(acc, bact, q, ks * JPK + jj, m);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// GROUPED = false: rows and both input tables through the index maps; the input block is 8 k-steps [A | B].
// GROUPED = true : rows (g, n) -> g * rows_per_group + n (the light-major rows of stage2/model/renderer.py:163,193 with
//   g = light, n = surface point), a workgroup never straddles two groups; the input block is table A only (4 k-steps)
//   and W_b * B[g] is part of the group's bias: one fp32 product per group and input layer (host side) instead of
//   4 k-steps per row and input layer -- 8 of 138 k-steps of the visibility network.
template <bool GROUPED>
__global__ __launch_bounds__(256, 1) void mlp_infer_bf16_kernel(Bf16Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsmem[];  // 2 x 72 KB weight stages
    constexpr int NIN = GROUPED ? 4 : 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int n_hidden = g.d.n_hidden;
    const unsigned group = GROUPED ? blockIdx.x / g.tiles_per_group : 0;
    const unsigned tile = GROUPED ? blockIdx.x - group * g.tiles_per_group : blockIdx.x;
    const unsigned char* gb = GROUPED ? g.gbias + (size_t)group * g.n_in_layers * kKsBytes : nullptr;  // this group's bias k-steps

    const unsigned char* wptr = g.w;  // source of the NEXT stage to request
    const unsigned char* bsrc = GROUPED ? gb : g.w + NIN * kKsBytes;  // ... and of its bias k-step
    int in_idx = 1;                   // input layers seen so far (layer 0 is one)
#pragma unroll
    for (int j = 0; j < 2 * NIN + 2; ++j) bf_dma_piece<2 * NIN>(wptr, bsmem, bsrc, NIN * kKsBytes, wave, lane, j);  // layer 0 = input block + bias
    wptr += NIN * kKsBytes + (GROUPED ? 0 : kKsBytes);

    // rows and table offsets (n_rows < 2^31 and tables < 4 GB are checked on the host: 32-bit index arithmetic)
    unsigned row[2], offa[2], offb[2];
    bool valid[2];
    const bool has_b = !GROUPED && g.tb != nullptr;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned n = tile * (unsigned)(kBfWaves * 64) + wave * 64 + t * 32 + ln;
        if constexpr (GROUPED) {
            valid[t] = n < g.rows_per_group;
            row[t] = group * g.rows_per_group + n;
            offa[t] = (valid[t] ? n : g.rows_per_group - 1) * 128u + lh * 16;
            offb[t] = 0;
        } else {
            valid[t] = n < g.n_rows;
            row[t] = n;
            const unsigned rc = valid[t] ? n : g.n_rows - 1;
            offa[t] = ((rc / g.a_div) % g.a_mod) * 128u + lh * 16;
            offb[t] = ((rc / g.b_div) % g.b_mod) * 128u + lh * 16;
        }
    }
    // input block: k-step s < 4 = features [16 s, 16 s + 16) of table A, s >= 4 of table B; the lane takes 8 h .. 8 h + 7.
    // (a missing table B re-reads A and is zeroed afterwards: no divergent load count)
    const unsigned char* tbp = has_b ? g.tb : g.ta;
    auto load_in = [&](bf16x8 (&bin)[2][8]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bin[t][s] = *reinterpret_cast<const bf16x8*>(g.ta + offa[t] + s * 32);
                if constexpr (!GROUPED) bin[t][4 + s] = *reinterpret_cast<const bf16x8*>(tbp + (has_b ? offb[t] : offa[t]) + s * 32);
            }
        }
    };
    // (mask_in also makes the compiler wait for the loads HERE, before the stage issues LDS-DMA pieces: those are asm
    // statements its vmcnt bookkeeping does not see, so a counted wait placed behind them would wait for them as well)
    auto mask_in = [&](bf16x8 (&bin)[2][8]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < NIN; ++s) asm volatile("" : "+v"(bin[t][s]));
        if (!GROUPED && !has_b) {
            const intx4 z = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 4; s < 8; ++s) bin[t][s] = __builtin_bit_cast(bf16x8, z);
        }
    };
    bf16x8 bias_b;  // K index 8 h + j: slots 0 and 1 carry the constant 1 (bias hi / lo columns)
    {
        intx4 o = {lh == 0 ? 0x3F803F80 : 0, 0, 0, 0};
        bias_b = __builtin_bit_cast(bf16x8, o);
    }

    floatx16 acc[2][8];
    bf16x8 bact[2][16];
    int gstage = 0;

    // One stage: wait for this wave's LDS-DMA pieces, barrier, then the MFMAs with the request for the next stage (NPS
    // stream pieces per wave from wptr, and, if NPT = NPS + 2, the wave's two pieces of its bias k-step from bsrc into the
    // slot behind its 8 activation k-steps) and the epilogue jobs in their gaps.  ADV = bytes the weight pointer advances
    // (the true stream size of the next stage; a request may over-read into the stage after it).
#define BF_STAGE(SRC, EPI, NPS, NPT, ADV)                                                                \
    {                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                                 \
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(bsmem + (gstage & 1) * kStageBytes);          \
        unsigned char* nxt = bsmem + ((gstage + 1) & 1) * kStageBytes;                                   \
        if ((EPI) == BF_EPI_FIRST) { /* pairs 2, 3 enter the stage in AGPRs: their reads then sit at the jobs, not at the top of the block */ \
            asm volatile("" : "+a"(acc[0][4]), "+a"(acc[1][4]), "+a"(acc[0][5]), "+a"(acc[1][5]),       \
                              "+a"(acc[0][6]), "+a"(acc[1][6]), "+a"(acc[0][7]), "+a"(acc[1][7]));      \
        }                                                                                                \
        bf_stage_mma<SRC, EPI, NIN, NPT>(acc, bact, bin, bias_b, wl, lane,                               \
                                         [&](int j_) { bf_dma_piece<NPS>(wptr, nxt, bsrc, 8 * kKsBytes, wave, lane, j_); }); \
        if ((EPI) == BF_EPI_FIRST) { /* k-steps 8..15 are first READ two stages on: without a use here hipcc sinks their epilogue there */ \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                            \
                asm volatile("" :: "v"(bact[t_][8]), "v"(bact[t_][9]), "v"(bact[t_][10]), "v"(bact[t_][11]), \
                                   "v"(bact[t_][12]), "v"(bact[t_][13]), "v"(bact[t_][14]), "v"(bact[t_][15])); \
        }                                                                                                \
        wptr += (ADV);                                                                                   \
        ++gstage;                                                                                        \
    }
    // source of the bias k-step of layer l's first stage (requested by the LAST stage of layer l - 1) and the stream
    // bytes of that stage: 8 activation k-steps, plus the bias k-step unless it comes from the group table
    auto next_first = [&](int l, int& adv) {
        const bool from_group = GROUPED && l < n_hidden && g.d.has_in[l] != 0;
        bsrc = from_group ? gb + in_idx * kKsBytes : wptr + 8 * kKsBytes;
        adv = l < n_hidden ? (from_group ? 8 : 9) * kKsBytes : 16 * 1024;  // (the final layer: 16 KB)
    };

    bf16x8 bin[2][8];
    int adv;
    // layer 0: the input block only (its own LAST stage; no previous layer)
    load_in(bin);
    mask_in(bin);
    next_first(1, adv);
    BF_STAGE(BF_SRC_INB, BF_EPI_LAST, 16, 18, adv)
    for (int li = 1; li < n_hidden; ++li) {
        const bool has_in = g.d.has_in[li] != 0;
        BF_STAGE(BF_SRC_ACT_LO, BF_EPI_FIRST, 16, 16, has_in ? NIN * kKsBytes : 8 * kKsBytes)
        if (has_in) {  // skip layer cat[y, x]: the input block goes in the middle, so that every layer ends with the same stage
            ++in_idx;
            load_in(bin);
            mask_in(bin);
            BF_STAGE(BF_SRC_IN, BF_EPI_NONE, 16, 16, 8 * kKsBytes)
        }
        next_first(li + 1, adv);
        BF_STAGE(BF_SRC_ACT_HI, BF_EPI_LAST, 16, 18, adv)
    }
    // the last hidden layer's pairs 2, 3 have no following stage to hide in
#pragma unroll
    for (int j = 0; j < 8; ++j) bf_epilogue_job(acc, bact, 2, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) bf_epilogue_job(acc, bact, 3, j);
#undef BF_STAGE
    // final layer: one output tile (n_out <= 32); two accumulator chains per row tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(bsmem + (gstage & 1) * kStageBytes);
        floatx16 f[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[t][c][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 a = wl[ks * 64 + lane];
            f[0][ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bact[0][ks], f[0][ks & 1], 0, 0, 0);
            f[1][ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bact[1][ks], f[1][ks & 1], 0, 0, 0);
        }
        const int n_out = g.d.n_out;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (valid[t]) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = 8 * (v >> 2) + 4 * lh + (v & 3);
                    if (m < n_out) {
                        float x = f[t][0][v] + f[t][1][v] + g.final_bias[m];
                        if (g.d.out_act == PSN_OUT_SIGMOID) x = sigmoidf_(x);
                        else if (g.d.out_act == PSN_OUT_OCC) x = sigmoidf_(x * -10.0f);
                        g.out[(int64_t)row[t] * n_out + m] = x;
                    }
                }
            }
        }
    }
}

// Bias k-steps of the row groups: V [n, 256] fp32 (one row per (group, input layer): W_b * B[group] + b) ->
// [n][8 output tiles][64 lanes][8] bf16 in the natural K order of a bias k-step: K slot 0 = bf16(v), slot 1 = bf16(v - slot 0).
__global__ __launch_bounds__(256) void bf16_pack_group_bias_kernel(const float* __restrict__ V, int64_t n, uint16_t* __restrict__ dst) {
    const int64_t total = n * 4096;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int ot = (int)((e >> 9) & 7);
        const int64_t r = e >> 12;
        const int k = 8 * (lane >> 5) + j;
        float v = 0.0f;
        if (k < 2) {
            const float b = V[r * 256 + 32 * ot + (lane & 31)];
            const float hi = (float)(__bf16)b;
            v = k == 0 ? hi : b - hi;
        }
        dst[e] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}

// W[rows, cols] (row-major, ldw floats per row), zero-extended, k-steps [ks0, ks0 + n_ks) -> [ks][ot][lane][8] bf16:
//   lane (m = lane & 31, h = lane >> 5), element j  <-  W[32 ot + m][k],
//   k = 16 ks + 8 h + j (natural order: input block, bias columns) or
//   k = 32 (ks / 2) + 16 (ks % 2) + 8 (j / 4) + 4 h + (j % 4) (permuted: previous layer's activations).
__global__ __launch_bounds__(256) void mlp_pack_bf16_kernel(const float* __restrict__ W, int64_t ldw, int rows, int cols,
                                                            int permuted, int n_ot, int ks0, int n_ks, uint16_t* __restrict__ dst) {
    const int64_t total = (int64_t)n_ks * n_ot * 512;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int64_t blk = e >> 9;
        const int ot = (int)(blk % n_ot);
        const int ks = ks0 + (int)(blk / n_ot);
        const int m = lane & 31, h = lane >> 5;
        const int r = 32 * ot + m;
        const int k = permuted ? 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * ks + 8 * h + j;
        float v = 0.0f;
        if (r < rows && k < cols) v = W[(int64_t)r * ldw + k];
        dst[e] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}

}  // namespace psn

extern "C" int psn_mlp_pack_bf16(const float* W, int64_t ldw, int rows, int cols, int permuted, int n_ot, int ks0, int n_ks,
                                 uint16_t* dst, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(W && dst, "mlp_pack_bf16: null pointer");
    PSN_CHECK_ARG((n_ot == 8 || n_ot == 1) && ks0 >= 0 && n_ks >= 1 && ks0 + n_ks <= 16, "mlp_pack_bf16: n_ot=%d ks0=%d n_ks=%d", n_ot, ks0, n_ks);
    PSN_CHECK_ARG(rows >= 1 && rows <= 32 * n_ot && cols >= 1 && cols <= 256 && ldw >= cols, "mlp_pack_bf16: %d x %d (ldw %lld) does not fit",
                  rows, cols, (long long)ldw);
    const int64_t total = (int64_t)n_ks * n_ot * 512;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mlp_pack_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, ldw, rows, cols, permuted, n_ot, ks0, n_ks, dst);
    PSN_CHECK_LAUNCH("mlp_pack_bf16");
    return PSN_OK;
}

static int bf16_launch(bool grouped, psn::Bf16Args& a, int64_t blocks, void* stream, const char* what) {
    using namespace psn;
    PSN_CHECK_ARG(blocks < (1ll << 31), "%s: too many rows", what);
    const size_t lds_bytes = 2 * kStageBytes;
    // 144 KB of dynamic LDS need the opt-in attribute; set per call (per-device state, cheap, no static flag to race on)
    const void* fn = grouped ? reinterpret_cast<const void*>(&mlp_infer_bf16_kernel<true>) : reinterpret_cast<const void*>(&mlp_infer_bf16_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %zu bytes of LDS: %s", what, lds_bytes, hipGetErrorString(e));
        return PSN_E_LAUNCH;
    }
    if (grouped) hipLaunchKernelGGL(mlp_infer_bf16_kernel<true>, dim3((unsigned)blocks), dim3(kBfWaves * 64), lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_infer_bf16_kernel<false>, dim3((unsigned)blocks), dim3(kBfWaves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH(what);
    return PSN_OK;
}

static int bf16_check_desc(const PsnBf16Desc& d, const char* what) {
    using namespace psn;
    PSN_CHECK_ARG(d.n_hidden >= 1 && d.n_hidden <= PSN_MLP_MAX_LAYERS, "%s: n_hidden=%d", what, d.n_hidden);
    PSN_CHECK_ARG(d.n_out >= 1 && d.n_out <= 32, "%s: n_out=%d", what, d.n_out);
    PSN_CHECK_ARG(d.out_act >= PSN_OUT_NONE && d.out_act <= PSN_OUT_OCC, "%s: out_act=%d", what, d.out_act);
    PSN_CHECK_ARG(d.has_in[0] != 0, "%s: layer 0 must read the input block", what);
    return PSN_OK;
}

extern "C" int psn_mlp_infer_bf16(const PsnBf16Desc* desc, const uint16_t* packed_w, const float* final_bias,
                                  const uint16_t* tab_a, int64_t a_div, int64_t a_mod, const uint16_t* tab_b, int64_t b_div,
                                  int64_t b_mod, int64_t n_rows, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && final_bias && tab_a && out, "mlp_infer_bf16: null pointer");
    const PsnBf16Desc& d = *desc;
    if (int rc = bf16_check_desc(d, "mlp_infer_bf16")) return rc;
    PSN_CHECK_ARG(a_div >= 1 && a_mod >= 1 && (tab_b == nullptr || (b_div >= 1 && b_mod >= 1)), "mlp_infer_bf16: bad index map");
    PSN_CHECK_ARG((((uintptr_t)packed_w | (uintptr_t)tab_a | (uintptr_t)tab_b) & 15) == 0, "mlp_infer_bf16: buffers must be 16-byte aligned");
    if (n_rows <= 0) return PSN_OK;
    Bf16Args a = {};
    a.d = d;
    a.w = reinterpret_cast<const unsigned char*>(packed_w);
    a.final_bias = final_bias;
    a.ta = reinterpret_cast<const unsigned char*>(tab_a);
    a.tb = reinterpret_cast<const unsigned char*>(tab_b);
    PSN_CHECK_ARG(n_rows < (1ll << 31) && a_div < (1ll << 31) && a_mod <= (1ll << 24) && b_div < (1ll << 31) && b_mod <= (1ll << 24),
                  "mlp_infer_bf16: 32-bit index arithmetic: n_rows, divisors < 2^31, table rows <= 2^24");
    a.a_div = (unsigned)a_div; a.a_mod = (unsigned)a_mod;
    a.b_div = (unsigned)(b_div > 0 ? b_div : 1); a.b_mod = (unsigned)(b_mod > 0 ? b_mod : 1);
    a.n_rows = (unsigned)n_rows;
    a.out = out;
    const int rows_per_block = kBfWaves * 64;
    return bf16_launch(false, a, (n_rows + rows_per_block - 1) / rows_per_block, stream, "mlp_infer_bf16");
}

extern "C" int psn_mlp_infer_bf16_grouped(const PsnBf16Desc* desc, const uint16_t* packed_w, const float* final_bias,
                                          const uint16_t* tab_a, int64_t rows_per_group, const uint16_t* group_bias,
                                          int64_t n_groups, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && final_bias && tab_a && group_bias && out, "mlp_infer_bf16_grouped: null pointer");
    const PsnBf16Desc& d = *desc;
    if (int rc = bf16_check_desc(d, "mlp_infer_bf16_grouped")) return rc;
    PSN_CHECK_ARG((((uintptr_t)packed_w | (uintptr_t)tab_a | (uintptr_t)group_bias) & 15) == 0, "mlp_infer_bf16_grouped: buffers must be 16-byte aligned");
    PSN_CHECK_ARG(rows_per_group >= 0 && n_groups >= 0 && rows_per_group <= (1ll << 24) && rows_per_group * n_groups < (1ll << 31),
                  "mlp_infer_bf16_grouped: 32-bit index arithmetic: rows per group <= 2^24, rows < 2^31");
    if (rows_per_group == 0 || n_groups == 0) return PSN_OK;
    Bf16Args a = {};
    a.d = d;
    a.w = reinterpret_cast<const unsigned char*>(packed_w);
    a.final_bias = final_bias;
    a.ta = reinterpret_cast<const unsigned char*>(tab_a);
    a.gbias = reinterpret_cast<const unsigned char*>(group_bias);
    a.rows_per_group = (unsigned)rows_per_group;
    a.tiles_per_group = (unsigned)((rows_per_group + kBfWaves * 64 - 1) / (kBfWaves * 64));
    a.n_rows = (unsigned)(rows_per_group * n_groups);
    int n_in = 0;
    for (int l = 0; l < d.n_hidden; ++l) n_in += d.has_in[l] != 0;
    a.n_in_layers = (unsigned)n_in;
    a.out = out;
    return bf16_launch(true, a, (int64_t)a.tiles_per_group * n_groups, stream, "mlp_infer_bf16_grouped");
}

extern "C" int psn_bf16_pack_group_bias(const float* V, int64_t n, uint16_t* dst, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(V && dst && n >= 1 && n < (1ll << 30), "bf16_pack_group_bias: V / dst / n=%lld", (long long)n);
    const int64_t total = n * 4096;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(bf16_pack_group_bias_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, n, dst);
    PSN_CHECK_LAUNCH("bf16_pack_group_bias");
    return PSN_OK;
}
