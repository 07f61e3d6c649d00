// bf16 inference engine for the 256-wide ReLU networks of stage 2 -- the evaluation / relighting path only
// (BASELINE config 5: "bf16 MFMA path ... envmap relight eval"; stage2/eval.py:199-218 evaluates visibility_net on
// 512 environment lights x every surface pixel, no gradients).  Training and every parity-gated path stay on the
// exact-fp32 engine of mlp_infer.hip; this kernel is opt-in (PSNetwork.inference_precision = 'bf16').
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
// of 8-9 k-steps (64 / 72 KB), double buffered.
//
// Roofline: MFMA-bound in bf16 (2.5 PFLOP/s dense): per layer and wave 256-272 MFMAs (32 cycles each) against
// ~640 VALU instructions of epilogue (accumulator reads, cvt, max).
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
};

constexpr int kKsBytes = 8192;               // one k-step of a hidden layer: 8 output tiles x 64 lanes x 16 B
constexpr int kStageBytes = 9 * kKsBytes;    // largest stage: 8 k-steps + the bias k-step
constexpr int kBfWaves = 4;

// LDS-DMA 4 x NPW KB of the weight stream: wave w moves the contiguous blocks [w NPW, (w + 1) NPW) of 1 KB each.  Four
// consecutive blocks share one base (global address and M0) and differ in the instruction offset only, so a stage costs
// NPW global_load_lds plus ~NPW/4 address / M0 updates, no branches: the scheduler can spread them between MFMAs.
template <int NPW>
__device__ __forceinline__ void bf_stage_dma(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, int wave, int lane) {
    const unsigned char* g = gsrc + wave * (NPW * 1024);
    unsigned char* l = lds_dst + wave * (NPW * 1024);
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int grp = j >> 2;
        const unsigned char* base = g + grp * 4096;
        auto gp = (const __attribute__((address_space(1))) void*)(base + (unsigned)(lane * 16));
        auto lp = (__attribute__((address_space(3))) void*)(l + grp * 4096);
        switch (j & 3) {
            case 0: __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0); break;
        }
    }
}

// NKS k-steps of one stage against the 8 output tiles, both row tiles sharing each weight fragment.  Fragments are
// double buffered per k-step (8 ds_read_b128 = 8 KB per wave, issued at the start of the previous k-step's 16 MFMAs).
template <int NKS, bool ZERO_C, int NPW, typename RequestNext>
__device__ __forceinline__ void bf_stage_mma(floatx16 (&acc)[2][8], const bf16x8 (&b0)[NKS], const bf16x8 (&b1)[NKS],
                                             const bf16x8* __restrict__ wl, int lane, RequestNext request_next) {
    bf16x8 a[2][8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) a[0][ot] = wl[ot * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);  // the first k-step's fragments are in flight ...
    request_next();                     // ... the NPW LDS-DMA requests are spread over the first MFMAs below
    floatx16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) {
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) a[(ks + 1) & 1][ot] = wl[((ks + 1) * 8 + ot) * 64 + lane];
        }
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
            acc[0][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][ot], b0[ks], (ZERO_C && ks == 0) ? zero : acc[0][ot], 0, 0, 0);
            acc[1][ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][ot], b1[ks], (ZERO_C && ks == 0) ? zero : acc[1][ot], 0, 0, 0);
        }
        // pin the order: per pair of MFMAs one fragment read of the next k-step and (k-steps 0..2) one LDS-DMA request
        constexpr int kPerKs = 8;
        const int dma_here = NPW - ks * kPerKs > kPerKs ? kPerKs : (NPW - ks * kPerKs > 0 ? NPW - ks * kPerKs : 0);
        if (ks + 1 < NKS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (k < dma_here) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    }
}

// ReLU + round-to-nearest-even bf16 of accumulator registers [8 qp, 8 qp + 8): the next layer's B operand.
__device__ __forceinline__ bf16x8 bf_pack_relu(const floatx16& c, int qp) {
    intx4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        floatx2 f;
        f[0] = c[8 * qp + 2 * i];
        f[1] = c[8 * qp + 2 * i + 1];
        shortx2 s = __builtin_bit_cast(shortx2, __builtin_convertvector(f, bf16x2));
        const shortx2 z = {0, 0};
        s = __builtin_elementwise_max(s, z);  // negative floats are negative int16: max with 0 is ReLU (and -0 -> +0)
        o[i] = __builtin_bit_cast(int, s);
    }
    return __builtin_bit_cast(bf16x8, o);
}

__global__ __launch_bounds__(256, 1) void mlp_infer_bf16_kernel(Bf16Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsmem[];  // 2 x 72 KB weight stages
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int n_hidden = g.d.n_hidden;

    const unsigned char* wptr = g.w;  // source of the NEXT stage to request
    bf_stage_dma<18>(wptr, bsmem, wave, lane);  // layer 0 = input block + bias (9 k-steps)
    wptr += 72 * 1024;

    // rows and table offsets (n_rows < 2^31 and tables < 4 GB are checked on the host: 32-bit index arithmetic)
    unsigned row[2], offa[2], offb[2];
    const bool has_b = g.tb != nullptr;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        row[t] = blockIdx.x * (unsigned)(kBfWaves * 64) + wave * 64 + t * 32 + ln;
        const unsigned rc = row[t] < g.n_rows ? row[t] : g.n_rows - 1;
        offa[t] = ((rc / g.a_div) % g.a_mod) * 128u + lh * 16;
        offb[t] = ((rc / g.b_div) % g.b_mod) * 128u + lh * 16;
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
                bin[t][4 + s] = *reinterpret_cast<const bf16x8*>(tbp + (has_b ? offb[t] : offa[t]) + s * 32);
            }
        }
    };
    auto mask_in = [&](bf16x8 (&bin)[2][8]) {
        if (!has_b) {
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

    // One stage: wait for this wave's LDS-DMA pieces, barrier, then the
    // MFMAs with the request for the next stage (4 x NPW KB; NPW = 0: none) spread between them.  ADV = bytes the
    // weight pointer advances (the true size of the next stage; a request may over-read into the stage after it).
#define BF_STAGE(NKS, ZERO, B0, B1, NPW, ADV)                                                            \
    {                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                                 \
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(bsmem + (gstage & 1) * kStageBytes);          \
        unsigned char* nxt = bsmem + ((gstage + 1) & 1) * kStageBytes;                                   \
        bf_stage_mma<NKS, ZERO, NPW>(acc, B0, B1, wl, lane, [&]() { bf_stage_dma<NPW>(wptr, nxt, wave, lane); }); \
        wptr += (ADV);                                                                                   \
        ++gstage;                                                                                        \
    }
#define BF_EPILOGUE()                                                             \
    {                                                                             \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                             \
            _Pragma("unroll") for (int ot = 0; ot < 8; ++ot) {                    \
                bact[t][2 * ot] = bf_pack_relu(acc[t][ot], 0);                    \
                bact[t][2 * ot + 1] = bf_pack_relu(acc[t][ot], 1);                \
            }                                                                     \
    }

    {  // layer 0: the input block only
        bf16x8 bin[2][8];
        load_in(bin);
        mask_in(bin);
        bf16x8 s0[9], s1[9];
#pragma unroll
        for (int s = 0; s < 8; ++s) { s0[s] = bin[0][s]; s1[s] = bin[1][s]; }
        s0[8] = bias_b; s1[8] = bias_b;
        BF_STAGE(9, true, s0, s1, 18, n_hidden > 1 ? 72 * 1024 : 16 * 1024)
        BF_EPILOGUE()
    }
    for (int li = 1; li < n_hidden; ++li) {
        const bool has_in = g.d.has_in[li] != 0;
        const int first_next = li + 1 < n_hidden ? 72 * 1024 : 16 * 1024;  // next layer's first stage / the final layer
        {
            bf16x8 s0[9], s1[9];
#pragma unroll
            for (int s = 0; s < 8; ++s) { s0[s] = bact[0][s]; s1[s] = bact[1][s]; }
            s0[8] = bias_b; s1[8] = bias_b;
            BF_STAGE(9, true, s0, s1, 16, 64 * 1024)
        }
        {
            bf16x8 s0[8], s1[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) { s0[s] = bact[0][8 + s]; s1[s] = bact[1][8 + s]; }
            BF_STAGE(8, false, s0, s1, 18, has_in ? 64 * 1024 : first_next)
        }
        if (has_in) {  // skip layer: cat[y, x] (fetching x earlier, under the stage above, costs more in spills than it hides)
            bf16x8 bin[2][8];
            load_in(bin);
            mask_in(bin);
            BF_STAGE(8, false, bin[0], bin[1], 18, first_next)
        }
        BF_EPILOGUE()
    }
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
            if (row[t] < g.n_rows) {
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
#undef BF_STAGE
#undef BF_EPILOGUE
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

extern "C" int psn_mlp_infer_bf16(const PsnBf16Desc* desc, const uint16_t* packed_w, const float* final_bias,
                                  const uint16_t* tab_a, int64_t a_div, int64_t a_mod, const uint16_t* tab_b, int64_t b_div,
                                  int64_t b_mod, int64_t n_rows, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && final_bias && tab_a && out, "mlp_infer_bf16: null pointer");
    const PsnBf16Desc& d = *desc;
    PSN_CHECK_ARG(d.n_hidden >= 1 && d.n_hidden <= PSN_MLP_MAX_LAYERS, "mlp_infer_bf16: n_hidden=%d", d.n_hidden);
    PSN_CHECK_ARG(d.n_out >= 1 && d.n_out <= 32, "mlp_infer_bf16: n_out=%d", d.n_out);
    PSN_CHECK_ARG(d.out_act >= PSN_OUT_NONE && d.out_act <= PSN_OUT_OCC, "mlp_infer_bf16: out_act=%d", d.out_act);
    PSN_CHECK_ARG(d.has_in[0] != 0, "mlp_infer_bf16: layer 0 must read the input block");
    PSN_CHECK_ARG(a_div >= 1 && a_mod >= 1 && (tab_b == nullptr || (b_div >= 1 && b_mod >= 1)), "mlp_infer_bf16: bad index map");
    PSN_CHECK_ARG((((uintptr_t)packed_w | (uintptr_t)tab_a | (uintptr_t)tab_b) & 15) == 0, "mlp_infer_bf16: buffers must be 16-byte aligned");
    if (n_rows <= 0) return PSN_OK;
    Bf16Args a;
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
    const int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    PSN_CHECK_ARG(blocks < (1ll << 31), "mlp_infer_bf16: too many rows");
    const size_t lds_bytes = 2 * kStageBytes;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_infer_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            set_error("mlp_infer_bf16: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
            return PSN_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(mlp_infer_bf16_kernel, dim3((unsigned)blocks), dim3(kBfWaves * 64), lds_bytes, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_infer_bf16");
    return PSN_OK;
}
