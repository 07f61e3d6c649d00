// Split-bf16 ("bf16x6") inference engine for the 256-wide ReLU networks of stage 2: fp32-class accuracy on the bf16 matrix
// pipe.  EXPERIMENT, opt-in (PSNetwork.inference_precision = 'bf16x6' / conf train.vis_bf16x6), gradient-free rows only: the
// L shading rows of stage2/model/renderer.py:191-200 (vis.detach() at :197) and the relighting evaluation of
// stage2/eval.py:199-218.  The exact-fp32 engine of mlp_infer.hip stays the default and the headline.
//
// Every fp32 operand is the exact sum of three bf16 numbers, x = x_hi + x_mid + x_lo (8 + 8 + 8 significant bits, each piece
// the round-to-nearest bf16 of what the previous ones left), and a product is evaluated as the six partial products whose
// weight is >= 2^-16 of the full one,
//     w x ~= w_hi x_hi + w_hi x_mid + w_mid x_hi + w_hi x_lo + w_mid x_mid + w_lo x_hi        (fp32 accumulation in the MFMA),
// dropping w_mid x_lo + w_lo x_mid + w_lo x_lo <= 3 * 2^-24 |w x|: one fp32 ulp per product, i.e. the error of an fp32 FMA
// chain.  bf16 MFMA runs at 16x the fp32 MFMA rate, so six of them still are 2.7x the fp32 peak on paper.
//
// Tiling (v_mfma_f32_32x32x16_bf16: M = 32 features, N = 32 rows, K = 16): a wave owns 32 rows x 256 features = 128 fp32
// accumulators and keeps the three planes of its activations (3 x 16 k-steps x 4 registers) in registers in the B-operand
// layout; the C/D registers of a layer, ReLU'd and split, ARE the next layer's B operands through the permuted K order of
// the bf16 engine (mlp_infer_bf16.hip: feature 32 ot + 16 qp + 8 (j / 4) + 4 h + (j % 4) <-> k-step 2 ot + qp, slot 8 h + j).
// Workgroup = 4 waves = 128 rows of ONE row group (light), one workgroup per CU (1 wave per SIMD, <= 512 registers).
// Weights stream L2 -> LDS by LDS-DMA in stages of 2 k-steps x 3 planes x 8 output tiles = 48 KB, double buffered; the bias
// of a layer (its three pieces in K slots 0..2 of one k-step, against the constant operand (1, 1, 1, 0, ...): exact) rides
// with the layer's first stage.  Input block: as in the exact-fp32 engine, a layer is linear in it, so W_in [pe(x_n) | pe(l_g)]
// + b = U[n] + V[g] with U = W_a pe(x) (one fp32 row per point) and V = W_b pe(l) + b (one per group) computed by two small
// fp32 GEMMs on the host side; the layers that read the input block START their accumulators from U[n] + V[g] (fp32 adds,
// the row of U prefetched under the previous layer's epilogue) and spend no MFMA on it: layer 0 is that sum alone.
#include <cstdlib>
#include "common.h"

namespace psn {

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef float xfloatx2 __attribute__((ext_vector_type(2)));
typedef int xintx4 __attribute__((ext_vector_type(4)));

struct X3Args {
    PsnBf16Desc d;
    const unsigned char* w;        // weight stream in execution order (48 KB stages)
    const unsigned char* bias;     // [n_hidden][8 KB] bias k-steps of the layers WITHOUT an input block
    const float* final_bias;
    const float* U;                // [rows_per_group][n_in_layers * 256] fp32: W_a pe(x_n) of every input layer
    const float* V;                // [n_groups][n_in_layers * 256] fp32: W_b pe(l_g) + b
    unsigned rows_per_group, tiles_per_group, n_in_layers;
    float* out;
    // OCC variant (psn_mlp_infer_x3_occ): rows are query points, encoded in the kernel
    const float* points;           // [n_rows][3]
    long long n_rows;              // rows (capacity when n_rows_dev is given)
    const long long* n_rows_dev;   // optional: the row count lives on the device
    const int64_t* out_rows;       // optional: row r's occupancy goes to out[out_rows[r]]
    int pe_octaves; float pe_scale;
    int skip_layer;                // hidden layer whose input is cat[h, pe] / sqrt(2) (stage1/model/network.py:90-91), -1: none
    int pe_first;                  // first input feature of that layer that is a positional-encoding column (217)
    // sweep form (psn_march_sweep_x3; stage1/model/rendering.py:447-462): the rows are (ray, step) pairs, a workgroup = 128
    // consecutive steps of ONE ray, workgroups in block-major order; the point is formed in the prologue (n_steps > 0)
    const float* ray_o;            // [n_rays][3]
    const float* ray_d;            // [n_rays][3]
    const float* far;              // [n_rays] sphere exit depth
    const float* u;                // [n_steps] linspace(0, 1, n_steps)
    const float* omu;              // [n_steps] 1 - u
    float near, tau;
    int n_steps;
    int* skip;                     // [n_rays] or nullptr: INT_MAX - b = block b of the ray holds a sign change (lowest b wins), 0 = none
    unsigned long long* n_blocks;  // optional: counts the evaluated 128-step blocks (measurement only)
};

constexpr int kX3Piece = 1024;
constexpr int kX3KsBytes = 24 * 1024;        // one k-step: 8 output tiles x 3 planes x 1 KB
constexpr int kX3StageBytes = 2 * kX3KsBytes;  // 48 KB
constexpr int kX3BiasBytes = 8 * 1024;
constexpr int kX3BufBytes = kX3StageBytes + kX3BiasBytes;  // 56 KB per LDS buffer
constexpr int kX3Waves = 4;
constexpr int kX3PeStride = 48;              // OCC: per-row encoding in LDS, 3 natural k-steps of 16 columns (39 real + zeros)
constexpr int kX3PeBytes = kX3Waves * 32 * kX3PeStride * 4;  // 24 KB behind the two stage buffers

// this wave's share of a stage request: pieces [12 wave, 12 wave + 12) of the 48 KB stream part, and -- with_bias -- pieces
// [2 wave, 2 wave + 2) of the layer's 8 KB bias k-step
__device__ __forceinline__ void x3_dma_piece(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, const unsigned char* __restrict__ bsrc,
                                             int wave, int lane, int j) {
    const unsigned voff = lane * 16;
    if (j < 12) {
        const int grp = j >> 2;
        const unsigned char* base = gsrc + wave * (12 * kX3Piece) + grp * 4096;
        const unsigned lds = lds_addr(lds_dst + wave * (12 * kX3Piece) + grp * 4096);
        switch (j & 3) {
            case 0: lds_dma_16<0>(base, lds, voff); break;
            case 1: lds_dma_16<1024>(base, lds, voff); break;
            case 2: lds_dma_16<2048>(base, lds, voff); break;
            default: lds_dma_16<3072>(base, lds, voff); break;
        }
    } else {
        const unsigned char* base = bsrc + wave * 2048;
        const unsigned lds = lds_addr(lds_dst + kX3StageBytes + wave * 2048);
        if (j == 12) lds_dma_16<0>(base, lds, voff);
        else lds_dma_16<1024>(base, lds, voff);
    }
}

__device__ __forceinline__ float x3_bf16_to_f32_lo(int packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float x3_bf16_to_f32_hi(int packed) { return __builtin_bit_cast(float, packed & (int)0xFFFF0000); }
__device__ __forceinline__ int x3_cvt2(float a, float b) {
    xfloatx2 f;
    f[0] = a; f[1] = b;
    return __builtin_bit_cast(int, __builtin_convertvector(f, xbf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
// two fp32 values -> their hi / mid / lo bf16 pairs (each piece exact in fp32: the residuals have <= 16 significant bits)
__device__ __forceinline__ void x3_split2(float a, float b, int& hi, int& mid, int& lo) {
    hi = x3_cvt2(a, b);
    const float ra = a - x3_bf16_to_f32_lo(hi), rb = b - x3_bf16_to_f32_hi(hi);
    mid = x3_cvt2(ra, rb);
    lo = x3_cvt2(ra - x3_bf16_to_f32_lo(mid), rb - x3_bf16_to_f32_hi(mid));
}

// The six partial products of one (k-step, output tile pair): fragments a[plane][o] (o = tile of the pair), B planes
// b[plane].  Consecutive MFMAs alternate between the two accumulators, smallest terms first.
__device__ __forceinline__ void x3_mma_pair(floatx16& c0, floatx16& c1, const xbf16x8 (&a)[3][2], const xbf16x8& bh, const xbf16x8& bm, const xbf16x8& bl) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][1], bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bh, c1, 0, 0, 0);
}

// One 48 KB stage = 2 k-steps against all 8 output tiles: 8 (k-step, tile pair) groups of 12 MFMAs; the 6 fragment reads of
// group g + 1 are issued at the head of group g, this wave's LDS-DMA pieces of the next stage ride in the first groups.
template <typename BOp, typename RequestPiece>
__device__ __forceinline__ void x3_stage_mma(floatx16 (&acc)[8], const xbf16x8* __restrict__ wl, int lane, int n_pieces, BOp bop, RequestPiece request_piece) {
    xbf16x8 a[2][3][2];
    auto load_frags = [&](int grp, xbf16x8 (&f)[3][2]) __attribute__((always_inline)) {
        const int ks = grp >> 2, p = grp & 3;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f[pl][o] = wl[((ks * 8 + 2 * p + o) * 3 + pl) * 64 + lane];
    };
    load_frags(0, a[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const int ks = grp >> 2, p = grp & 3;
        if (grp + 1 < 8) {
            load_frags(grp + 1, a[(grp + 1) & 1]);
            // the reads are ISSUED here, a whole group (12 MFMAs = 384 cycles) ahead of their use: without the region boundary
            // the register allocator merges the two fragment sets and sinks the reads behind this group's last MFMAs, and with
            // one wave per SIMD their latency is then exposed at the head of every group
            __builtin_amdgcn_sched_barrier(0);
        }
        x3_mma_pair(acc[2 * p], acc[2 * p + 1], a[grp & 1], bop(0, ks), bop(1, ks), bop(2, ks));
        // the next stage's pieces go out in the first two groups: a stage lasts only 96 MFMAs = 3072 cycles, and a piece issued
        // in its second half lands after the stage has ended (an L2 round trip is ~1500 cycles)
        if (grp < 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) request_piece(6 * grp + j);
        } else if (grp == 2 && n_pieces > 12) { request_piece(12); request_piece(13); }  // the bias k-step of the next layer (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// OCC = false: the grouped ReLU network of stage 2 (input block through the fp32 init tables U / V, see the header comment).
// OCC = true (psn_mlp_infer_x3_occ): the stage-1 occupancy network (stage1/model/network.py:85-101) on query points -- the
// positional encoding is formed in the prologue with the expressions of pe_encode / mlp_infer_kernel<SRC = 2> and parked in LDS
// (48 floats per row); layer 0 multiplies its three planes in natural K order (two 48 KB stages: k-steps 0..3, the fourth all
// zeros); every layer starts from its bias k-step; the activation is softplus(beta = 100) (common.h softplus100_pair, the
// exact-fp32 engine's code) before the split; the epilogue IN FRONT of the skip layer writes the encoding columns into the
// input features >= pe_first (217 ... 255: cat[h, pe] of network.py:90, the 1 / sqrt(2) is folded into the packed weights),
// so that layer is an ordinary 256-input layer; output = sigmoid(-10 logit), optionally scattered (out_rows) for a row count
// that lives on the device (n_rows_dev) -- the shadow-ray path of stage1/model/rendering.py:378-408.
template <bool OCC>
__global__ __launch_bounds__(256, 1) void mlp_infer_x3_kernel(X3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];  // 2 x (48 KB stage + 8 KB bias k-step) [+ 24 KB encoding]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int n_hidden = g.d.n_hidden;
    unsigned group = 0, tile = blockIdx.x;
    long long n_rows_eff = 0;
    if constexpr (OCC) {
        n_rows_eff = g.n_rows;
        if (g.n_rows_dev != nullptr) { const long long nd = *g.n_rows_dev; n_rows_eff = nd < n_rows_eff ? nd : n_rows_eff; }
        if ((long long)blockIdx.x * (kX3Waves * 32) >= n_rows_eff) return;  // (wave-uniform: before anything is requested)
    } else {
        group = blockIdx.x / g.tiles_per_group;
        tile = blockIdx.x - group * g.tiles_per_group;
    }
    const unsigned init_stride = g.n_in_layers * 256u;
    const float* vrow = OCC ? nullptr : g.V + (size_t)group * init_stride + 4 * lh;

    const unsigned char* wptr = g.w;  // the NEXT stage to request
    int in_idx = 0;                   // input layers seen so far
    int gstage = 0;
    // OCC = false: first stage of layer 1 (layer 0 has no weights of its own left) + that layer's bias k-step unless it reads
    // the input block.  OCC = true: first stage of layer 0 + its bias k-step.
    const bool l1_bias = OCC ? true : (n_hidden > 1 && g.d.has_in[1] == 0);
    const unsigned char* first_bias = OCC ? g.bias : g.bias + kX3BiasBytes;
#pragma unroll
    for (int j = 0; j < 12; ++j) x3_dma_piece(wptr, xsmem, first_bias, wave, lane, j);
    if (l1_bias) { x3_dma_piece(wptr, xsmem, first_bias, wave, lane, 12); x3_dma_piece(wptr, xsmem, first_bias, wave, lane, 13); }
    wptr += kX3StageBytes;

    const unsigned n = tile * (unsigned)(kX3Waves * 32) + wave * 32 + ln;
    const bool valid = OCC ? (long long)n < n_rows_eff : n < g.rows_per_group;
    const unsigned row = OCC ? n : group * g.rows_per_group + n;
    const float* urow = OCC ? nullptr : g.U + (size_t)(valid ? n : g.rows_per_group - 1) * init_stride + 4 * lh;
    float* pe_row = reinterpret_cast<float*>(xsmem + 2 * kX3BufBytes) + (wave * 32 + ln) * kX3PeStride;  // OCC: this lane's row
    if constexpr (OCC) {
        // positional encoding of this lane's point (network.py:141-150): [p s, sin(2^f p s), cos(2^f p s)] per octave f; the
        // two lanes of a row share the 18 (octave, coordinate) sincos pairs; expressions as compute_xin_tile (mlp_infer.hip)
        float q[3] = {0.f, 0.f, 0.f};
        if (valid) { q[0] = g.points[(size_t)n * 3]; q[1] = g.points[(size_t)n * 3 + 1]; q[2] = g.points[(size_t)n * 3 + 2]; }
        const int n_pairs = 3 * g.pe_octaves, half = (n_pairs + 1) / 2;
        for (int pi = lh * half; pi < (lh == 0 ? half : n_pairs); ++pi) {
            const int f = pi / 3, c = pi - 3 * f;
            const float arg = ldexpf((c == 0 ? q[0] : (c == 1 ? q[1] : q[2])) * g.pe_scale, f);
            float sn, cs;
            sincosf(arg, &sn, &cs);
            pe_row[3 + 6 * f + c] = sn;
            pe_row[3 + 6 * f + 3 + c] = cs;
        }
        if (lh == 0) { pe_row[0] = q[0] * g.pe_scale; pe_row[1] = q[1] * g.pe_scale; pe_row[2] = q[2] * g.pe_scale; }
        else for (int col = 3 + 2 * n_pairs; col < kX3PeStride; ++col) pe_row[col] = 0.0f;
        // (read back by the same wave only -- layer 0 and the epilogue in front of the skip layer --, behind the barrier of
        //  the first stage)
    }
    xbf16x8 ones_b;  // K slots 0..2 (lane half 0) carry the constant 1: the three bias pieces add up exactly
    {
        xintx4 o = {lh == 0 ? 0x3F803F80 : 0, lh == 0 ? 0x00003F80 : 0, 0, 0};
        ones_b = __builtin_bit_cast(xbf16x8, o);
    }

    floatx16 acc[8];
    xbf16x8 bact[3][16];
    // lane (n, h), tile ot, register v = 4 q + r  <->  feature 32 ot + 8 q + 4 h + r: four consecutive floats per (ot, q).
    // acc = U[n] + V[g] (fp32).  (Requesting the U row ahead of the preceding epilogue needs 128 more registers than the
    // 512 a wave has -- measured: scratch spills --, so its latency is exposed once per input layer: ~1 % of a pass.)
    auto init_acc_uv = [&](int idx) __attribute__((always_inline)) {
        const float* pu = urow + idx * 256;
        const float* pv = vrow + idx * 256;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 u = *reinterpret_cast<const float4*>(pu + 32 * ot + 8 * q);
                const float4 t = *reinterpret_cast<const float4*>(pv + 32 * ot + 8 * q);
                acc[ot][4 * q] = u.x + t.x; acc[ot][4 * q + 1] = u.y + t.y; acc[ot][4 * q + 2] = u.z + t.z; acc[ot][4 * q + 3] = u.w + t.w;
            }
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) asm volatile("" : "+v"(acc[ot]));  // complete before the next LDS-DMA pieces are issued (asm: invisible to vmcnt)
    };

    // One stage: this wave's pieces have landed, barrier, MFMAs with the request for the next stage in their gaps.
#define X3_STAGE(BOP, NEXT_HAS_BIAS, BSRC)                                                                   \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        __syncthreads();                                                                                     \
        const xbf16x8* wl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes);            \
        unsigned char* nxt = xsmem + ((gstage + 1) & 1) * kX3BufBytes;                                       \
        const unsigned char* bsrc_ = (BSRC);                                                                 \
        x3_stage_mma(acc, wl, lane, (NEXT_HAS_BIAS) ? 14 : 12, BOP,                                          \
                     [&](int j_) { x3_dma_piece(wptr, nxt, bsrc_, wave, lane, j_); });                       \
        wptr += kX3StageBytes;                                                                               \
        ++gstage;                                                                                            \
    }
    // bias of the layer whose first stage sits in LDS buffer (gstage & 1): acc = b_hi + b_mid + b_lo (exact)
    auto init_acc_bias = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const xbf16x8* bl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes + kX3StageBytes);
        floatx16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ot * 64 + lane], ones_b, zero, 0, 0, 0);
    };
    // ReLU (OCC: softplus) + split of the finished accumulators into the three B-operand planes of the next layer; OCC,
    // inject: the features >= pe_first take the row's encoding columns instead (the cat[h, pe] in front of the skip layer;
    // pe_first >= 192, so only output tiles 6 and 7 are concerned)
    auto epilogue = [&](bool inject) __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                xintx4 oh, om, ol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float c0, c1;
                    if constexpr (OCC) {
                        f32x2 sp, unused;
                        softplus100_pair<false>(f32x2{acc[ot][8 * qp + 2 * i], acc[ot][8 * qp + 2 * i + 1]}, sp, unused);
                        c0 = sp.x; c1 = sp.y;
                        if (ot >= 6 && inject) {
                            const int f0 = 32 * ot + 16 * qp + 8 * ((2 * i) >> 2) + 4 * lh + ((2 * i) & 3);  // feature of slot j = 2 i
                            const int k0 = f0 - g.pe_first, k1 = f0 + 1 - g.pe_first;
                            const float p0 = pe_row[k0 < 0 ? 0 : k0], p1 = pe_row[k1 < 0 ? 0 : k1];
                            c0 = k0 >= 0 ? p0 : c0;
                            c1 = k1 >= 0 ? p1 : c1;
                        }
                    } else {
                        c0 = relu1(acc[ot][8 * qp + 2 * i]); c1 = relu1(acc[ot][8 * qp + 2 * i + 1]);
                    }
                    int h_, m_, l_;
                    x3_split2(c0, c1, h_, m_, l_);
                    oh[i] = h_; om[i] = m_; ol[i] = l_;
                }
                bact[0][2 * ot + qp] = __builtin_bit_cast(xbf16x8, oh);
                bact[1][2 * ot + qp] = __builtin_bit_cast(xbf16x8, om);
                bact[2][2 * ot + qp] = __builtin_bit_cast(xbf16x8, ol);
            }
        }
    };

    if constexpr (OCC) {
        // layer 0: bias + W_0 pe(x) on the matrix pipe, natural K order: k-step ks, slot j of lane (n, h) = column 16 ks + 8 h + j
        init_acc_bias();  // (its barrier also orders the encoding writes above against the reads below)
        xbf16x8 bin[3][4];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const float4 v0 = *reinterpret_cast<const float4*>(pe_row + 16 * ks + 8 * lh), v1 = *reinterpret_cast<const float4*>(pe_row + 16 * ks + 8 * lh + 4);
            xintx4 oh, om, ol;
            int h_, m_, l_;
            x3_split2(v0.x, v0.y, h_, m_, l_); oh[0] = h_; om[0] = m_; ol[0] = l_;
            x3_split2(v0.z, v0.w, h_, m_, l_); oh[1] = h_; om[1] = m_; ol[1] = l_;
            x3_split2(v1.x, v1.y, h_, m_, l_); oh[2] = h_; om[2] = m_; ol[2] = l_;
            x3_split2(v1.z, v1.w, h_, m_, l_); oh[3] = h_; om[3] = m_; ol[3] = l_;
            bin[0][ks] = __builtin_bit_cast(xbf16x8, oh); bin[1][ks] = __builtin_bit_cast(xbf16x8, om); bin[2][ks] = __builtin_bit_cast(xbf16x8, ol);
        }
        {
            xintx4 z = {0, 0, 0, 0};
            bin[0][3] = bin[1][3] = bin[2][3] = __builtin_bit_cast(xbf16x8, z);
        }
        X3_STAGE(([&](int pl, int ks) -> xbf16x8 { return bin[pl][ks]; }), false, g.bias)
        X3_STAGE(([&](int pl, int ks) -> xbf16x8 { return bin[pl][2 + ks]; }), n_hidden > 1, g.bias + kX3BiasBytes)
    } else {
        // layer 0: U[n] + V[g] alone
        init_acc_uv(0);
        ++in_idx;
    }
    for (int li = 1; li < n_hidden; ++li) {
        const bool has_in = !OCC && g.d.has_in[li] != 0;
        epilogue(OCC && li == g.skip_layer);                   // of layer li - 1
        if (has_in) { init_acc_uv(in_idx); ++in_idx; }
        else init_acc_bias();
        const bool next_bias = li + 1 < n_hidden && (OCC || g.d.has_in[li + 1] == 0);
        const unsigned char* next_bsrc = g.bias + (size_t)(li + 1 < n_hidden ? li + 1 : 0) * kX3BiasBytes;
#define X3_ACT_STAGE(S, LAST)                                                                                 \
        X3_STAGE(([&](int pl, int ks) -> xbf16x8 { return bact[pl][2 * (S) + ks]; }), (LAST) && next_bias, next_bsrc)
        X3_ACT_STAGE(0, false) X3_ACT_STAGE(1, false) X3_ACT_STAGE(2, false) X3_ACT_STAGE(3, false)
        X3_ACT_STAGE(4, false) X3_ACT_STAGE(5, false) X3_ACT_STAGE(6, false) X3_ACT_STAGE(7, true)
#undef X3_ACT_STAGE
    }
    epilogue(false);
#undef X3_STAGE
    // final layer: one output tile (n_out <= 32), 16 k-steps x 3 planes in ONE 48 KB stage; four accumulator chains
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const xbf16x8* wl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes);
        floatx16 f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) f[c][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const xbf16x8 ah = wl[(ks * 3 + 0) * 64 + lane], am = wl[(ks * 3 + 1) * 64 + lane], al = wl[(ks * 3 + 2) * 64 + lane];
            floatx16& c = f[ks & 3];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bact[0][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[2][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bact[1][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bact[0][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[1][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[0][ks], c, 0, 0, 0);
        }
        const int n_out = g.d.n_out;
        if (valid) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = 8 * (v >> 2) + 4 * lh + (v & 3);
                if (m < n_out) {
                    float x = (f[0][v] + f[1][v]) + (f[2][v] + f[3][v]) + g.final_bias[m];
                    if (g.d.out_act == PSN_OUT_SIGMOID) x = sigmoidf_(x);
                    else if (g.d.out_act == PSN_OUT_OCC) x = sigmoidf_(x * -10.0f);
                    const int64_t orow = (OCC && g.out_rows != nullptr) ? g.out_rows[row] : (int64_t)row;
                    g.out[orow * n_out + m] = x;
                }
            }
        }
    }
}

// ============================================================================================================================
// PIPELINED form (round 4): the epilogue of a layer -- activation + three-way split of 128 accumulators per lane, ~10 k cycles of
// vector work next to 24.6 k cycles of MFMAs per layer, fully exposed with one wave per SIMD -- runs INSIDE the next layer's MFMA
// stream.  Stage S of layer l + 1 (k-steps 2 S, 2 S + 1) consumes exactly the features of OUTPUT TILE S of layer l, so only tile
// 0 has to be ready when the layer starts; the 8 (k-step, tile pair) groups of stage S each carry one eighth of the epilogue of
// tile S + 1 (one activation pair + its split, ~35 vector instructions in the gaps of 12 MFMAs).  That needs the previous
// layer's accumulators alive while the new ones accumulate: TWO accumulator sets (256 registers) -- paid for by keeping only
// TWO tiles of B-operand planes (48 registers instead of 192: a tile is dead once its stage has run).  The last stage of a layer
// writes its results back into the first set (MFMA with D != C), so the layer loop carries one set and needs no copies.
// 456 -> ~400 registers, one wave per SIMD as before.
template <bool OCC>
__device__ __forceinline__ void x3_epi_job(const floatx16& a, const int j, const int ot, const int lh, const bool inject, const float* pe_row,
                                           const int pe_first, xintx4 (&st)[3][2]) {
    const int qp = j >> 2, i = j & 3;
    float c0, c1;
    if constexpr (OCC) {
        f32x2 sp, unused;
        softplus100_pair<false>(f32x2{a[8 * qp + 2 * i], a[8 * qp + 2 * i + 1]}, sp, unused);
        c0 = sp.x; c1 = sp.y;
        if (ot >= 6 && inject) {
            const int f0 = 32 * ot + 16 * qp + 8 * ((2 * i) >> 2) + 4 * lh + ((2 * i) & 3);  // feature of slot j = 2 i
            const int k0 = f0 - pe_first, k1 = f0 + 1 - pe_first;
            const float p0 = pe_row[k0 < 0 ? 0 : k0], p1 = pe_row[k1 < 0 ? 0 : k1];
            c0 = k0 >= 0 ? p0 : c0;
            c1 = k1 >= 0 ? p1 : c1;
        }
    } else {
        c0 = relu1(a[8 * qp + 2 * i]); c1 = relu1(a[8 * qp + 2 * i + 1]);
    }
    int h_, m_, l_;
    x3_split2(c0, c1, h_, m_, l_);
    st[0][qp][i] = h_; st[1][qp][i] = m_; st[2][qp][i] = l_;
}

// x3_mma_pair whose FIRST product of each chain reads its C operand from (s0, s1) and writes (c0, c1): moves the accumulators
__device__ __forceinline__ void x3_mma_pair_move(floatx16& c0, floatx16& c1, const floatx16& s0, const floatx16& s1, const xbf16x8 (&a)[3][2],
                                                 const xbf16x8& bh, const xbf16x8& bm, const xbf16x8& bl) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][0], bh, s0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][1], bh, s1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bh, c1, 0, 0, 0);
}

// The epilogue of ONE accumulator pair (x3_epi_job) cut into six phases of <= 8 vector instructions (two quarter-rate
// transcendentals count as eight), one behind each PAIR of MFMAs of a group: the wave issues in order and a 32 x 32 x 16 MFMA
// holds the matrix pipe for 32 cycles, so <= 32 cycles of vector work between two MFMAs cost nothing, while the same work as
// one block behind a group's 12 MFMAs (what hipcc makes of a job handed over whole, with or without sched_group_barrier) runs
// after the pipe has drained.  Arithmetic: softplus100_pair<false> (common.h) and x3_split2, operation by operation.
template <bool OCC>
struct X3EpiPhases {
    f32x2 z, nat, hi, r, u, w, lw, l, sp;
    int h_, m_;
    __device__ __forceinline__ void run(const int k, const floatx16& a, const int j, const int ot, const int lh, const bool inject,
                                        const float* pe_row, const int pe_first, xintx4 (&st)[3][2]) {
        const int qp = j >> 2, i = j & 3;
        if constexpr (OCC) {
            const float L2E = 1.44269502162933349609375f, LN2_HI = 0.693147182464599609375f, LN2_LO = -1.904654323148236e-9f;
            const float C_HI = 0.00999999977648258209228515625f, C_LO = 2.2351741811588166e-10f;
            if (k == 0) {
                z = f32x2{a[8 * qp + 2 * i], a[8 * qp + 2 * i + 1]};
                const f32x2 t = z * 100.0f;
                nat = f32x2{fminf(t.x, -t.x), fminf(t.y, -t.y)};
                hi = nat * L2E;
            } else if (k == 1) {
                r = pk_fma(-hi, pk2(LN2_HI), nat);
                r = pk_fma(-hi, pk2(LN2_LO), r);
                u = f32x2{__builtin_amdgcn_exp2f(hi.x), __builtin_amdgcn_exp2f(hi.y)};
            } else if (k == 2) {
                u = pk_fma(u, r, u);
                w = u + 1.0f;
                lw = f32x2{__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)};
            } else if (k == 3) {
                const f32x2 c = w - 1.0f;
                const f32x2 d = u - c;
                const f32x2 rw = pk_fma(c, pk_fma(c, pk2(0.5f), pk2(-1.0f)), pk2(1.0f));
                l = pk_fma(lw, pk2(LN2_HI), d * rw);
            } else if (k == 4) {
                const f32x2 zr = {relu1(z.x), relu1(z.y)};
                sp = pk_fma(l, pk2(C_HI), pk_fma(l, pk2(C_LO), zr));
                if (ot >= 6 && inject) {
                    const int f0 = 32 * ot + 16 * qp + 8 * ((2 * i) >> 2) + 4 * lh + ((2 * i) & 3);
                    const int k0 = f0 - pe_first, k1 = f0 + 1 - pe_first;
                    const float p0 = pe_row[k0 < 0 ? 0 : k0], p1 = pe_row[k1 < 0 ? 0 : k1];
                    sp.x = k0 >= 0 ? p0 : sp.x;
                    sp.y = k1 >= 0 ? p1 : sp.y;
                }
                h_ = x3_cvt2(sp.x, sp.y);
                st[0][qp][i] = h_;
            } else {
                const float ra = sp.x - x3_bf16_to_f32_lo(h_), rb = sp.y - x3_bf16_to_f32_hi(h_);
                m_ = x3_cvt2(ra, rb);
                st[1][qp][i] = m_;
                st[2][qp][i] = x3_cvt2(ra - x3_bf16_to_f32_lo(m_), rb - x3_bf16_to_f32_hi(m_));
            }
        } else {
            if (k == 0) {
                sp = f32x2{relu1(a[8 * qp + 2 * i]), relu1(a[8 * qp + 2 * i + 1])};
                h_ = x3_cvt2(sp.x, sp.y);
                st[0][qp][i] = h_;
            } else if (k == 1) {
                z = f32x2{sp.x - x3_bf16_to_f32_lo(h_), sp.y - x3_bf16_to_f32_hi(h_)};
                m_ = x3_cvt2(z.x, z.y);
                st[1][qp][i] = m_;
            } else if (k == 2) {
                st[2][qp][i] = x3_cvt2(z.x - x3_bf16_to_f32_lo(m_), z.y - x3_bf16_to_f32_hi(m_));
            }
        }
    }
};

// One stage of the pipelined form.  dst / src: the accumulators written / read (the same set, except in a layer's last stage,
// whose first k-step moves src -> dst).  HAS_JOB: group grp carries accumulator pair grp of tile `job_ot` of the previous
// layer (job_acc) through its six phases, one behind each pair of MFMAs (the order of the 12 products is x3_mma_pair's).
template <bool OCC, bool MOVE, bool HAS_JOB, typename BOp, typename RequestPiece>
__device__ __forceinline__ void x3_stage_mma_p(floatx16 (&dst)[8], floatx16 (&src)[8], const xbf16x8* __restrict__ wl, int lane, int n_pieces, BOp bop,
                                               RequestPiece request_piece, const floatx16& job_acc, const int job_ot, const int lh, const bool inject,
                                               const float* pe_row, const int pe_first, xintx4 (&job_st)[3][2]) {
    xbf16x8 a[2][3][2];
    auto load_frags = [&](int grp, xbf16x8 (&f)[3][2]) __attribute__((always_inline)) {
        const int ks = grp >> 2, p = grp & 3;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f[pl][o] = wl[((ks * 8 + 2 * p + o) * 3 + pl) * 64 + lane];
    };
    load_frags(0, a[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const int ks = grp >> 2, p = grp & 3;
        if (grp + 1 < 8) {
            load_frags(grp + 1, a[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // (see x3_stage_mma: the reads are issued a whole group ahead of their use)
        }
        const xbf16x8 (&fr)[3][2] = a[grp & 1];
        const xbf16x8 bh = bop(0, ks), bm = bop(1, ks), bl = bop(2, ks);
        floatx16& c0 = dst[2 * p];
        floatx16& c1 = dst[2 * p + 1];
        X3EpiPhases<OCC> ph;
#define X3P_SUB(K, PL, B)                                                                                              \
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[PL][0], B, (MOVE && ks == 0 && (K) == 0) ? src[2 * p] : c0, 0, 0, 0);      \
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[PL][1], B, (MOVE && ks == 0 && (K) == 0) ? src[2 * p + 1] : c1, 0, 0, 0);  \
        if constexpr (HAS_JOB) { ph.run(K, job_acc, grp, job_ot, lh, inject, pe_row, pe_first, job_st); __builtin_amdgcn_sched_barrier(0); }
        X3P_SUB(0, 2, bh) X3P_SUB(1, 0, bl) X3P_SUB(2, 1, bm) X3P_SUB(3, 1, bh) X3P_SUB(4, 0, bm) X3P_SUB(5, 0, bh)
#undef X3P_SUB
        if (grp < 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) request_piece(6 * grp + j);
        } else if (grp == 2 && n_pieces > 12) { request_piece(12); request_piece(13); }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool OCC>
__global__ __launch_bounds__(256, 1) void mlp_infer_x3p_kernel(X3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];  // 2 x (48 KB stage + 8 KB bias k-step) [+ 24 KB encoding]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int n_hidden = g.d.n_hidden;
    unsigned group = 0, tile = blockIdx.x;
    long long n_rows_eff = 0;
    const bool sweep = OCC && g.n_steps > 0;
    long long m_ray = 0;
    int m_blk = 0;
    if constexpr (OCC) {
        n_rows_eff = g.n_rows;
        if (sweep) {
            // (ray, block) of this workgroup, block-major; a block BEHIND the lowest flagged block of its ray leaves here -- the flag was
            // raised by a workgroup that may have run on another XCD (agent scope), see mlp_infer_kernel<.., 3>
            const long long n_rays = g.n_rows / g.n_steps;
            m_ray = (long long)blockIdx.x % n_rays;
            m_blk = (int)((long long)blockIdx.x / n_rays);
            if (m_blk > 0 && g.skip != nullptr) {
                const int s = __hip_atomic_load(g.skip + m_ray, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s != 0 && 0x7fffffff - s < m_blk) return;
            }
            if (g.n_blocks != nullptr && threadIdx.x == 0) atomicAdd(g.n_blocks, 1ull);
        } else {
            if (g.n_rows_dev != nullptr) { const long long nd = *g.n_rows_dev; n_rows_eff = nd < n_rows_eff ? nd : n_rows_eff; }
            if ((long long)blockIdx.x * (kX3Waves * 32) >= n_rows_eff) return;
        }
    } else {
        group = blockIdx.x / g.tiles_per_group;
        tile = blockIdx.x - group * g.tiles_per_group;
    }
    const unsigned init_stride = g.n_in_layers * 256u;
    const float* vrow = OCC ? nullptr : g.V + (size_t)group * init_stride + 4 * lh;
    const unsigned char* wptr = g.w;
    int in_idx = 0, gstage = 0;
    const bool l1_bias = OCC ? true : (n_hidden > 1 && g.d.has_in[1] == 0);
    const unsigned char* first_bias = OCC ? g.bias : g.bias + kX3BiasBytes;
#pragma unroll
    for (int j = 0; j < 12; ++j) x3_dma_piece(wptr, xsmem, first_bias, wave, lane, j);
    if (l1_bias) { x3_dma_piece(wptr, xsmem, first_bias, wave, lane, 12); x3_dma_piece(wptr, xsmem, first_bias, wave, lane, 13); }
    wptr += kX3StageBytes;

    const int m_step = m_blk * (kX3Waves * 32) + wave * 32 + ln;  // sweep: the step of this lane's row
    const unsigned n = sweep ? (unsigned)(m_ray * g.n_steps + m_step) : tile * (unsigned)(kX3Waves * 32) + wave * 32 + ln;
    const bool valid = OCC ? (long long)n < n_rows_eff : n < g.rows_per_group;
    const unsigned row = OCC ? n : group * g.rows_per_group + n;
    const float* urow = OCC ? nullptr : g.U + (size_t)(valid ? n : g.rows_per_group - 1) * init_stride + 4 * lh;
    float* pe_row = reinterpret_cast<float*>(xsmem + 2 * kX3BufBytes) + (wave * 32 + ln) * kX3PeStride;
    if constexpr (OCC) {
        float q[3] = {0.f, 0.f, 0.f};
        if (sweep) {
            // sample_points_kernel (csrc/sample.hip), miss profile: d = near (1 - u) + far u, p = origin + dir d; products and sums
            // rounded separately (-ffp-contract=off): the point has the bits of the table the two-launch path writes
            const float d = g.near * g.omu[m_step] + g.far[m_ray] * g.u[m_step];
            q[0] = g.ray_o[m_ray * 3 + 0] + g.ray_d[m_ray * 3 + 0] * d;
            q[1] = g.ray_o[m_ray * 3 + 1] + g.ray_d[m_ray * 3 + 1] * d;
            q[2] = g.ray_o[m_ray * 3 + 2] + g.ray_d[m_ray * 3 + 2] * d;
        } else if (valid) { q[0] = g.points[(size_t)n * 3]; q[1] = g.points[(size_t)n * 3 + 1]; q[2] = g.points[(size_t)n * 3 + 2]; }
        const int n_pairs = 3 * g.pe_octaves, half = (n_pairs + 1) / 2;
        for (int pi = lh * half; pi < (lh == 0 ? half : n_pairs); ++pi) {
            const int f = pi / 3, c = pi - 3 * f;
            const float arg = ldexpf((c == 0 ? q[0] : (c == 1 ? q[1] : q[2])) * g.pe_scale, f);
            float sn, cs;
            sincosf(arg, &sn, &cs);
            pe_row[3 + 6 * f + c] = sn;
            pe_row[3 + 6 * f + 3 + c] = cs;
        }
        if (lh == 0) { pe_row[0] = q[0] * g.pe_scale; pe_row[1] = q[1] * g.pe_scale; pe_row[2] = q[2] * g.pe_scale; }
        else for (int col = 3 + 2 * n_pairs; col < kX3PeStride; ++col) pe_row[col] = 0.0f;
    }
    xbf16x8 ones_b;
    {
        xintx4 o = {lh == 0 ? 0x3F803F80 : 0, lh == 0 ? 0x00003F80 : 0, 0, 0};
        ones_b = __builtin_bit_cast(xbf16x8, o);
    }

    floatx16 cur[8], tmp[8];  // cur: the finished pre-activations of the previous layer (loop-carried); tmp: the layer being accumulated
    xintx4 st[2][3][2];       // B-operand planes of two output tiles: [ring slot][plane][k-step of the tile]

    auto init_uv = [&](floatx16 (&acc)[8], int idx) __attribute__((always_inline)) {
        const float* pu = urow + idx * 256;
        const float* pv = vrow + idx * 256;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 u = *reinterpret_cast<const float4*>(pu + 32 * ot + 8 * q);
                const float4 t = *reinterpret_cast<const float4*>(pv + 32 * ot + 8 * q);
                acc[ot][4 * q] = u.x + t.x; acc[ot][4 * q + 1] = u.y + t.y; acc[ot][4 * q + 2] = u.z + t.z; acc[ot][4 * q + 3] = u.w + t.w;
            }
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) asm volatile("" : "+v"(acc[ot]));
    };
    auto init_bias = [&](floatx16 (&acc)[8]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const xbf16x8* bl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes + kX3StageBytes);
        floatx16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ot * 64 + lane], ones_b, zero, 0, 0, 0);
    };

#define X3P_STAGE(MOVE, HAS_JOB, DST, SRC, BOP, NEXT_HAS_BIAS, BSRC, JOB_ACC, JOB_OT, JOB_ST)                  \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        __syncthreads();                                                                                     \
        const xbf16x8* wl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes);            \
        unsigned char* nxt = xsmem + ((gstage + 1) & 1) * kX3BufBytes;                                       \
        const unsigned char* bsrc_ = (BSRC);                                                                 \
        x3_stage_mma_p<OCC, MOVE, HAS_JOB>(DST, SRC, wl, lane, (NEXT_HAS_BIAS) ? 14 : 12, BOP,               \
                                           [&](int j_) { x3_dma_piece(wptr, nxt, bsrc_, wave, lane, j_); },  \
                                           JOB_ACC, JOB_OT, lh, inject_, pe_row, g.pe_first, JOB_ST);         \
        wptr += kX3StageBytes;                                                                               \
        ++gstage;                                                                                            \
    }

    bool inject_ = false;
    if constexpr (OCC) {
        init_bias(cur);
        xbf16x8 bin[3][4];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const float4 v0 = *reinterpret_cast<const float4*>(pe_row + 16 * ks + 8 * lh), v1 = *reinterpret_cast<const float4*>(pe_row + 16 * ks + 8 * lh + 4);
            xintx4 oh, om, ol;
            int h_, m_, l_;
            x3_split2(v0.x, v0.y, h_, m_, l_); oh[0] = h_; om[0] = m_; ol[0] = l_;
            x3_split2(v0.z, v0.w, h_, m_, l_); oh[1] = h_; om[1] = m_; ol[1] = l_;
            x3_split2(v1.x, v1.y, h_, m_, l_); oh[2] = h_; om[2] = m_; ol[2] = l_;
            x3_split2(v1.z, v1.w, h_, m_, l_); oh[3] = h_; om[3] = m_; ol[3] = l_;
            bin[0][ks] = __builtin_bit_cast(xbf16x8, oh); bin[1][ks] = __builtin_bit_cast(xbf16x8, om); bin[2][ks] = __builtin_bit_cast(xbf16x8, ol);
        }
        {
            xintx4 z = {0, 0, 0, 0};
            bin[0][3] = bin[1][3] = bin[2][3] = __builtin_bit_cast(xbf16x8, z);
        }
        X3P_STAGE(false, false, cur, cur, ([&](int pl, int ks) -> xbf16x8 { return bin[pl][ks]; }), false, g.bias, cur[0], 0, st[0])
        X3P_STAGE(false, false, cur, cur, ([&](int pl, int ks) -> xbf16x8 { return bin[pl][2 + ks]; }), n_hidden > 1, g.bias + kX3BiasBytes, cur[0], 0, st[0])
    } else {
        init_uv(cur, 0);
        ++in_idx;
    }
    for (int li = 1; li < n_hidden; ++li) {
        const bool has_in = !OCC && g.d.has_in[li] != 0;
        const bool inject = OCC && li == g.skip_layer;
        inject_ = inject;
        // tile 0 of the previous layer: the only part of its epilogue that nothing hides
#pragma unroll
        for (int j = 0; j < 8; ++j) x3_epi_job<OCC>(cur[0], j, 0, lh, inject, pe_row, g.pe_first, st[0]);
        if (has_in) { init_uv(tmp, in_idx); ++in_idx; }
        else init_bias(tmp);
        const bool next_bias = li + 1 < n_hidden && (OCC || g.d.has_in[li + 1] == 0);
        const unsigned char* next_bsrc = g.bias + (size_t)(li + 1 < n_hidden ? li + 1 : 0) * kX3BiasBytes;
        // stage S: MFMAs on tile S's planes (ring slot S & 1), the epilogue of tile S + 1 (into the other slot) in their gaps
#define X3P_ACT(S)                                                                                                           \
        X3P_STAGE(false, true, tmp, tmp, ([&](int pl, int ks) -> xbf16x8 { return __builtin_bit_cast(xbf16x8, st[(S) & 1][pl][ks]); }),   \
                  false, next_bsrc, cur[(S) + 1], (S) + 1, st[((S) + 1) & 1])
        X3P_ACT(0) X3P_ACT(1) X3P_ACT(2) X3P_ACT(3) X3P_ACT(4) X3P_ACT(5) X3P_ACT(6)
#undef X3P_ACT
        // last stage: every tile of the previous layer has been consumed -- the results move back into `cur`
        X3P_STAGE(true, false, cur, tmp, ([&](int pl, int ks) -> xbf16x8 { return __builtin_bit_cast(xbf16x8, st[1][pl][ks]); }),
                  next_bias, next_bsrc, cur[0], 0, st[0])
    }
#undef X3P_STAGE
    // final layer: the whole epilogue of the last hidden layer (tmp is dead: its registers hold the 16 k-steps of planes),
    // then one output tile (n_out <= 32), 16 k-steps x 3 planes in ONE 48 KB stage; four accumulator chains
    xbf16x8 bact[3][16];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        xintx4 s8[3][2];
#pragma unroll
        for (int j = 0; j < 8; ++j) x3_epi_job<OCC>(cur[ot], j, ot, lh, false, pe_row, g.pe_first, s8);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { bact[pl][2 * ot] = __builtin_bit_cast(xbf16x8, s8[pl][0]); bact[pl][2 * ot + 1] = __builtin_bit_cast(xbf16x8, s8[pl][1]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const xbf16x8* wl = reinterpret_cast<const xbf16x8*>(xsmem + (gstage & 1) * kX3BufBytes);
        floatx16 f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) f[c][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const xbf16x8 ah = wl[(ks * 3 + 0) * 64 + lane], am = wl[(ks * 3 + 1) * 64 + lane], al = wl[(ks * 3 + 2) * 64 + lane];
            floatx16& c = f[ks & 3];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bact[0][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[2][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bact[1][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bact[0][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[1][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bact[0][ks], c, 0, 0, 0);
        }
        const int n_out = g.d.n_out;
        float occ0 = 0.0f;  // output 0 of this lane's row (lanes of the lower half)
        if (valid) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = 8 * (v >> 2) + 4 * lh + (v & 3);
                if (m < n_out) {
                    float x = (f[0][v] + f[1][v]) + (f[2][v] + f[3][v]) + g.final_bias[m];
                    if (g.d.out_act == PSN_OUT_SIGMOID) x = sigmoidf_(x);
                    else if (g.d.out_act == PSN_OUT_OCC) x = sigmoidf_(x * -10.0f);
                    const int64_t orow = (OCC && g.out_rows != nullptr) ? g.out_rows[row] : (int64_t)row;
                    g.out[orow * n_out + m] = x;
                    if (v == 0) occ0 = x;
                }
            }
        }
        if constexpr (OCC) {
            if (sweep && g.skip != nullptr) {
                // val_m = occupancy - tau of the block's 128 steps through LDS (the encoding area: its rows were consumed by the skip
                // layer long ago); a negative product of neighbours, or a ray that does not start in free space, decides the ray:
                // its later blocks need not be evaluated (the semantics of mlp_infer_kernel<.., 3>, with 128-step blocks)
                float* xch = reinterpret_cast<float*>(xsmem + 2 * kX3BufBytes);
                __syncthreads();
                if (lh == 0) xch[wave * 32 + ln] = occ0 - g.tau;
                __syncthreads();
                if (tid < kX3Waves * 32 - 1) {
                    const bool hit = (xch[tid] * xch[tid + 1] < 0.0f) || (tid == 0 && m_blk == 0 && !(xch[0] < 0.0f));
                    if (hit) __hip_atomic_fetch_max(g.skip + m_ray, 0x7fffffff - m_blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ---- packers --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void x3_split1(float v, uint16_t (&p)[3]) {
    float r = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const __bf16 b = (__bf16)r;
        p[i] = __builtin_bit_cast(uint16_t, b);
        r -= (float)b;
    }
}
// W[rows, cols] (row-major, ldw floats per row), zero-extended, k-steps [ks0, ks0 + n_ks) -> [ks][ot][plane][lane][8] bf16;
// K order as psn_mlp_pack_bf16 (natural: k = 16 ks + 8 h + j; permuted: k = 32 (ks / 2) + 16 (ks % 2) + 8 (j / 4) + 4 h + (j % 4))
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ W, int64_t ldw, int rows, int cols, int permuted, int n_ot,
                                                      int ks0, int n_ks, uint16_t* __restrict__ dst) {
    const int64_t total = (int64_t)n_ks * n_ot * 512;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int64_t blk = e >> 9;
        const int ot = (int)(blk % n_ot);
        const int ksl = (int)(blk / n_ot);
        const int ks = ks0 + ksl;
        const int m = lane & 31, h = lane >> 5;
        const int r = 32 * ot + m;
        const int k = permuted ? 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * ks + 8 * h + j;
        float v = 0.0f;
        if (r < rows && k < cols) v = W[(int64_t)r * ldw + k];
        uint16_t p[3];
        x3_split1(v, p);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[(((int64_t)ksl * n_ot + ot) * 3 + pl) * 512 + lane * 8 + j] = p[pl];
    }
}
// V [n, 256] fp32 -> bias k-steps [n][8 tiles][64 lanes][8] bf16: K slots 0, 1, 2 (lane half 0) = hi, mid, lo of the value
__global__ __launch_bounds__(256) void x3_pack_bias_kernel(const float* __restrict__ V, int64_t n, uint16_t* __restrict__ dst) {
    const int64_t total = n * 4096;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int ot = (int)((e >> 9) & 7);
        const int64_t r = e >> 12;
        const int k = 8 * (lane >> 5) + j;
        uint16_t o = 0;
        if (k < 3) {
            uint16_t p[3];
            x3_split1(V[r * 256 + 32 * ot + (lane & 31)], p);
            o = p[k];
        }
        dst[e] = o;
    }
}
}  // namespace psn

static bool x3_pipelined() {
    const char* e = getenv("PSN_X3_PIPE");
    return !(e != nullptr && e[0] == '0');
}

extern "C" int psn_x3_pack(const float* W, int64_t ldw, int rows, int cols, int permuted, int n_ot, int ks0, int n_ks, uint16_t* dst,
                           void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(W && dst, "x3_pack: null pointer");
    PSN_CHECK_ARG((n_ot == 8 || n_ot == 1) && ks0 >= 0 && n_ks >= 1 && ks0 + n_ks <= 16, "x3_pack: n_ot=%d ks0=%d n_ks=%d", n_ot, ks0, n_ks);
    PSN_CHECK_ARG(rows >= 1 && rows <= 32 * n_ot && cols >= 1 && cols <= 256 && ldw >= cols, "x3_pack: %d x %d (ldw %lld) does not fit", rows, cols, (long long)ldw);
    const int64_t total = (int64_t)n_ks * n_ot * 512;
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, rows, cols, permuted, n_ot, ks0, n_ks, dst);
    PSN_CHECK_LAUNCH("x3_pack");
    return PSN_OK;
}

extern "C" int psn_x3_pack_bias(const float* V, int64_t n, uint16_t* dst, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(V && dst && n >= 1 && n < (1ll << 30), "x3_pack_bias: V / dst / n=%lld", (long long)n);
    const int64_t total = n * 4096;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(x3_pack_bias_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, n, dst);
    PSN_CHECK_LAUNCH("x3_pack_bias");
    return PSN_OK;
}

extern "C" int psn_mlp_infer_x3_grouped(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                                        const float* U, int64_t rows_per_group, const float* V, int64_t n_groups, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && bias_steps && final_bias && U && V && out, "mlp_infer_x3_grouped: null pointer");
    const PsnBf16Desc& d = *desc;
    PSN_CHECK_ARG(d.n_hidden >= 2 && d.n_hidden <= PSN_MLP_MAX_LAYERS && d.n_out >= 1 && d.n_out <= 32, "mlp_infer_x3_grouped: n_hidden=%d n_out=%d", d.n_hidden, d.n_out);
    PSN_CHECK_ARG(d.out_act >= PSN_OUT_NONE && d.out_act <= PSN_OUT_OCC && d.has_in[0] != 0, "mlp_infer_x3_grouped: out_act=%d, layer 0 must read the input block", d.out_act);
    PSN_CHECK_ARG((((uintptr_t)packed_w | (uintptr_t)bias_steps | (uintptr_t)U | (uintptr_t)V) & 15) == 0, "mlp_infer_x3_grouped: buffers must be 16-byte aligned");
    PSN_CHECK_ARG(rows_per_group >= 0 && n_groups >= 0 && rows_per_group <= (1ll << 24) && rows_per_group * n_groups < (1ll << 31),
                  "mlp_infer_x3_grouped: 32-bit index arithmetic: rows per group <= 2^24, rows < 2^31");
    if (rows_per_group == 0 || n_groups == 0) return PSN_OK;
    X3Args a = {};
    a.d = d;
    a.w = reinterpret_cast<const unsigned char*>(packed_w);
    a.bias = reinterpret_cast<const unsigned char*>(bias_steps);
    a.final_bias = final_bias;
    a.U = U; a.V = V;
    a.rows_per_group = (unsigned)rows_per_group;
    a.tiles_per_group = (unsigned)((rows_per_group + kX3Waves * 32 - 1) / (kX3Waves * 32));
    int n_in = 0;
    for (int l = 0; l < d.n_hidden; ++l) n_in += d.has_in[l] != 0;
    a.n_in_layers = (unsigned)n_in;
    a.out = out;
    const int64_t blocks = (int64_t)a.tiles_per_group * n_groups;
    PSN_CHECK_ARG(blocks < (1ll << 31), "mlp_infer_x3_grouped: too many rows");
    const size_t lds_bytes = 2 * kX3BufBytes;
    // PSN_X3_PIPE=0 selects the round-3 form (whole epilogue between two layers) for A/B runs; default: the pipelined form
    const auto kern = x3_pipelined() ? &mlp_infer_x3p_kernel<false> : &mlp_infer_x3_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        set_error("mlp_infer_x3_grouped: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return PSN_E_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kX3Waves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_infer_x3_grouped");
    return PSN_OK;
}

// Stage-1 occupancy network on the split-bf16 engine: sigmoid(-10 logit) of n_rows query points, see mlp_infer_x3_kernel<true>.
extern "C" int psn_mlp_infer_x3_occ(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                                    const float* points, int64_t n_rows, const long long* n_rows_dev, const int64_t* out_rows,
                                    int pe_octaves, float pe_scale, int skip_layer, int pe_first, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && bias_steps && final_bias && points && out, "mlp_infer_x3_occ: null pointer");
    const PsnBf16Desc& d = *desc;
    PSN_CHECK_ARG(d.n_hidden >= 2 && d.n_hidden <= PSN_MLP_MAX_LAYERS && d.n_out >= 1 && d.n_out <= 32, "mlp_infer_x3_occ: n_hidden=%d n_out=%d", d.n_hidden, d.n_out);
    PSN_CHECK_ARG(d.out_act >= PSN_OUT_NONE && d.out_act <= PSN_OUT_OCC, "mlp_infer_x3_occ: out_act=%d", d.out_act);
    PSN_CHECK_ARG(pe_octaves >= 0 && 3 + 6 * pe_octaves <= kX3PeStride, "mlp_infer_x3_occ: %d octaves do not fit %d encoding columns", pe_octaves, kX3PeStride);
    PSN_CHECK_ARG(skip_layer < d.n_hidden && (skip_layer < 1 || (pe_first >= 192 && pe_first + 3 + 6 * pe_octaves <= 256)),
                  "mlp_infer_x3_occ: skip_layer=%d pe_first=%d (the encoding columns must be input features 192..255 of a layer >= 1)", skip_layer, pe_first);
    PSN_CHECK_ARG((((uintptr_t)packed_w | (uintptr_t)bias_steps) & 15) == 0, "mlp_infer_x3_occ: buffers must be 16-byte aligned");
    PSN_CHECK_ARG(n_rows >= 0 && n_rows < (1ll << 31), "mlp_infer_x3_occ: 32-bit row arithmetic: rows < 2^31");
    if (n_rows == 0) return PSN_OK;
    X3Args a = {};
    a.d = d;
    a.w = reinterpret_cast<const unsigned char*>(packed_w);
    a.bias = reinterpret_cast<const unsigned char*>(bias_steps);
    a.final_bias = final_bias;
    a.points = points; a.n_rows = n_rows; a.n_rows_dev = n_rows_dev; a.out_rows = out_rows;
    a.pe_octaves = pe_octaves; a.pe_scale = pe_scale; a.skip_layer = skip_layer < 1 ? -1 : skip_layer; a.pe_first = pe_first;
    a.out = out;
    const int64_t blocks = (n_rows + kX3Waves * 32 - 1) / (kX3Waves * 32);
    const size_t lds_bytes = 2 * kX3BufBytes + kX3PeBytes;
    const auto kern = x3_pipelined() ? &mlp_infer_x3p_kernel<true> : &mlp_infer_x3_kernel<true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        set_error("mlp_infer_x3_occ: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return PSN_E_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kX3Waves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_infer_x3_occ");
    return PSN_OK;
}

// The ray-march sweep of stage1/model/rendering.py:447-462 on the split-bf16 engine (opt-in experiment; psn_march_sweep is the
// exact-fp32 form): occ [n_rays, n_steps] = sigmoid(-10 logit) of the points origin + dir (near (1 - u_m) + far u_m), formed and
// encoded in the kernel; a workgroup = 128 consecutive steps of one ray, block-major; skip [n_rays] int32 (zeroed by the caller,
// or NULL: every block is evaluated): the blocks behind a ray's first sign change are left out (their entries stay unwritten --
// nothing reads them: psn_first_crossing stops at the first crossing).  n_steps a multiple of 128.
extern "C" int psn_march_sweep_x3(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                                  const float* origin, const float* dir, const float* far, const float* u, const float* omu, float near,
                                  int64_t n_rays, int n_steps, float tau, int pe_octaves, float pe_scale, int skip_layer, int pe_first,
                                  int* skip, float* occ, unsigned long long* n_blocks, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && bias_steps && final_bias && origin && dir && far && u && omu && occ, "march_sweep_x3: null pointer");
    const PsnBf16Desc& d = *desc;
    PSN_CHECK_ARG(d.n_hidden >= 2 && d.n_hidden <= PSN_MLP_MAX_LAYERS && d.n_out == 1 && d.out_act == PSN_OUT_OCC, "march_sweep_x3: the occupancy network (one output, OCC)");
    PSN_CHECK_ARG(pe_octaves >= 0 && 3 + 6 * pe_octaves <= kX3PeStride, "march_sweep_x3: %d octaves do not fit %d encoding columns", pe_octaves, kX3PeStride);
    PSN_CHECK_ARG(skip_layer < d.n_hidden && (skip_layer < 1 || (pe_first >= 192 && pe_first + 3 + 6 * pe_octaves <= 256)), "march_sweep_x3: skip_layer=%d pe_first=%d", skip_layer, pe_first);
    PSN_CHECK_ARG(n_steps >= 128 && n_steps % (kX3Waves * 32) == 0 && n_rays >= 0 && n_rays * (int64_t)n_steps < (1ll << 31),
                  "march_sweep_x3: n_steps=%d must be a multiple of 128, rays x steps < 2^31", n_steps);
    PSN_CHECK_ARG(x3_pipelined(), "march_sweep_x3: built for the pipelined kernel only (PSN_X3_PIPE=0 is set)");
    if (n_rays == 0) return PSN_OK;
    X3Args a = {};
    a.d = d;
    a.w = reinterpret_cast<const unsigned char*>(packed_w);
    a.bias = reinterpret_cast<const unsigned char*>(bias_steps);
    a.final_bias = final_bias;
    a.n_rows = n_rays * (int64_t)n_steps;
    a.pe_octaves = pe_octaves; a.pe_scale = pe_scale; a.skip_layer = skip_layer < 1 ? -1 : skip_layer; a.pe_first = pe_first;
    a.out = occ;
    a.ray_o = origin; a.ray_d = dir; a.far = far; a.u = u; a.omu = omu; a.near = near; a.tau = tau; a.n_steps = n_steps;
    a.skip = skip; a.n_blocks = n_blocks;
    const int64_t blocks = n_rays * (n_steps / (kX3Waves * 32));
    const size_t lds_bytes = 2 * kX3BufBytes + kX3PeBytes;
    const auto kern = &mlp_infer_x3p_kernel<true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        set_error("march_sweep_x3: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return PSN_E_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kX3Waves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("march_sweep_x3");
    return PSN_OK;
}
