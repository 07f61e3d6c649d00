// Fully fused MLP inference for 256-wide networks: activations never leave registers.
//
// Replaces the no-grad network evaluations of the reference:
//   stage2 visibility_net over L*Ns rows   (stage2/model/renderer.py:191-200; vis.detach() at :197)
//   stage1 occupancy-only queries          (stage1/model/rendering.py:456-462 march, :537-540 secant,
//                                           :394-399 light_visibility; network.py:124-125)
//
// Formulation: per layer OUT^T[features, points] = W[features, K] * ACT^T[K, points] on
// v_mfma_f32_32x32x2_f32.  One wave owns 32 points (the MFMA N dimension = lane & 31); a lane holds
// 128 of the 256 features of its point in registers, in exactly the MFMA C/D layout
//     feature(mt, r, h) = 32*mt + (r & 3) + 8*(r >> 2) + 4*h,    h = lane >> 5.
// Because K may be visited in any order as long as A and B agree, the D registers of layer l are fed
// straight back as the B operand of layer l+1 (k-step (kt, r) consumes register act[kt][r], whose two
// lane halves hold features 32kt+(r&3)+8(r>>2)+{0,4}); the weights are pre-packed to match
// (psn_mlp_pack_layer), so there is no transpose, no LDS round trip and no HBM traffic for
// activations.  Weights (L2-resident, <= 2.6 MB) stream through LDS in 32 KB stages by LDS-DMA
// (global_load_lds_dwordx4), double buffered, shared by the 4 waves of the workgroup.
//
// Roofline: MFMA-bound.  Per stage and wave: 128 MFMAs (8192 cycles) vs 32 ds_read_b128.
#include "common.h"

namespace psn {

struct InferArgs {
    PsnMlpDesc d;
    const float* w;
    const float* b;
    const float* ta;
    int64_t a_div, a_mod;
    const float* tb;
    int64_t b_div, b_mod;
    int64_t n_rows;
    float* out;
};

constexpr int kStageFloats = 8 * 4 * 64 * 4;  // 8 m-tiles x 4 rho x 64 lanes x float4 = 32 KB

// One k-tile (32 input features = 16 MFMA k-steps) against NMT output tiles.
template <int NMT>
__device__ __forceinline__ void stage_compute(floatx16 (&acc)[8], const floatx16& bsrc, const float4* __restrict__ wl,
                                              int lane) {
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        float4 a[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) a[mt] = wl[(rho * NMT + mt) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bsrc[4 * rho + 0], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bsrc[4 * rho + 1], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bsrc[4 * rho + 2], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bsrc[4 * rho + 3], acc[mt], 0, 0, 0);
    }
}

// LDS-DMA one stage (n_blocks x 1 KB) of packed weights; the 4 waves split the blocks.
__device__ __forceinline__ void stage_load(const float* __restrict__ gsrc, float* lds_dst, int n_blocks, int wave,
                                           int lane) {
    for (int blk = wave; blk < n_blocks; blk += 4) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + blk * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds_dst + blk * 256), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256, 1) void mlp_infer_kernel(InferArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x 32 KB weight stages
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + lj;
    const int64_t rowc = row < g.n_rows ? row : g.n_rows - 1;

    // ---- schedule bookkeeping: flat list of stages over all layers -------------------------------
    const int n_layers = g.d.n_layers;
    // prefetch stage 0 of layer 0
    {
        const PsnMlpLayer& L0 = g.d.layers[0];
        stage_load(g.w + L0.w_off, smem, 4 * L0.n_mt, wave, lane);
    }

    // ---- input features -> registers (MFMA B-operand layout) ------------------------------------
    floatx16 xin[4];
    {
        const int64_t ia = (rowc / g.a_div) % g.a_mod;
        const float* pa = g.ta + ia * (int64_t)(g.d.in_kt_a * 32);
        const float* pb = nullptr;
        if (g.d.in_kt_b > 0) {
            const int64_t ib = (rowc / g.b_div) % g.b_mod;
            pb = g.tb + ib * (int64_t)(g.d.in_kt_b * 32);
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float* src = nullptr;
            if (kt < g.d.in_kt_a) src = pa + kt * 32;
            else if (kt < g.d.in_kt_a + g.d.in_kt_b) src = pb + (kt - g.d.in_kt_a) * 32;
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src != nullptr) v = *reinterpret_cast<const float4*>(src + 8 * rho + 4 * lh);
                xin[kt][4 * rho + 0] = v.x;
                xin[kt][4 * rho + 1] = v.y;
                xin[kt][4 * rho + 2] = v.z;
                xin[kt][4 * rho + 3] = v.w;
            }
        }
    }

    floatx16 act[8];
    floatx16 acc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[mt][r] = 0.f;

    // Hidden layers (8 output tiles) run in this loop; the final layer (1 output tile) is peeled off
    // below so that the accumulators never meet a control-flow merge between differently shaped code
    // paths (which makes hipcc shuttle all 128 accumulator registers between VGPRs and AGPRs per stage).
    int gstage = 0;  // global stage counter -> LDS buffer parity

#define PSN_STAGE(NMT, BSRC, S_IDX)                                                                         \
    {                                                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's LDS-DMA pieces have landed */        \
        __syncthreads(); /* every wave's pieces landed; the other buffer is no longer being read */         \
        const int s_ = (S_IDX);                                                                             \
        float* nxt = smem + ((gstage + 1) & 1) * kStageFloats;                                              \
        if (s_ + 1 < n_st) {                                                                                \
            stage_load(wl_g + (int64_t)(s_ + 1) * stage_floats, nxt, 4 * (NMT), wave, lane);                \
        } else if (li + 1 < n_layers) {                                                                     \
            const PsnMlpLayer& Ln = g.d.layers[li + 1];                                                     \
            stage_load(g.w + Ln.w_off, nxt, 4 * Ln.n_mt, wave, lane);                                       \
        }                                                                                                   \
        const float4* wl = reinterpret_cast<const float4*>(smem + (gstage & 1) * kStageFloats);             \
        stage_compute<NMT>(acc, BSRC, wl, lane);                                                            \
        ++gstage;                                                                                           \
    }

    int li = 0;
    for (; li < n_layers - 1; ++li) {
        const PsnMlpLayer L = g.d.layers[li];
        const int n_st = L.n_kt_in + L.n_kt_act;
        const int stage_floats = 8 * 1024;
        const float* wl_g = g.w + L.w_off;
        {  // bias -> accumulator init
            const float* bp = g.b + L.b_off;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 bv = *reinterpret_cast<const float4*>(bp + mt * 32 + 8 * q + 4 * lh);
                    acc[mt][4 * q + 0] = bv.x;
                    acc[mt][4 * q + 1] = bv.y;
                    acc[mt][4 * q + 2] = bv.z;
                    acc[mt][4 * q + 3] = bv.w;
                }
            }
        }
        // K tiles from the input features first, then from the previous activations (matches the packer)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < L.n_kt_in) PSN_STAGE(8, xin[kt], kt)
        }
        if (L.n_kt_act > 0) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) PSN_STAGE(8, act[kt], L.n_kt_in + kt)
        }
        // activation: accumulators become the next layer's B operands
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float z = acc[mt][r];
                float a;
                if (L.act == PSN_ACT_RELU) a = fmaxf(z, 0.0f);
                else if (L.act == PSN_ACT_SOFTPLUS100) a = softplus100(z);
                else a = z;
                act[mt][r] = a;
            }
        }
    }
    {   // final layer: one output tile
        const PsnMlpLayer L = g.d.layers[li];
        const int n_st = L.n_kt_in + L.n_kt_act;
        const int stage_floats = 1024;
        const float* wl_g = g.w + L.w_off;
        const float* bp = g.b + L.b_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q + 4 * lh);
            acc[0][4 * q + 0] = bv.x;
            acc[0][4 * q + 1] = bv.y;
            acc[0][4 * q + 2] = bv.z;
            acc[0][4 * q + 3] = bv.w;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < L.n_kt_in) PSN_STAGE(1, xin[kt], kt)
        }
        if (L.n_kt_act > 0) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) PSN_STAGE(1, act[kt], L.n_kt_in + kt)
        }
    }
#undef PSN_STAGE

    // ---- output: final layer has one m-tile; feature f = (r&3) + 8*(r>>2) + 4*h ------------------
    if (row < g.n_rows) {
        const int n_out = g.d.n_out;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (f < n_out) {
                float v = acc[0][r];
                if (g.d.out_act == PSN_OUT_SIGMOID) v = sigmoidf_(v);
                else if (g.d.out_act == PSN_OUT_OCC) v = sigmoidf_(v * -10.0f);
                g.out[row * n_out + f] = v;
            }
        }
    }
}

// Dense zero-padded W[n_mt*32][k_tiles*32] -> stage order [kt][rho][mt][lane][4]:
//   lane (i = lane & 31, h = lane >> 5), component c  <-  W[32*mt + i][32*kt + 8*rho + 4*h + c]
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ W, int64_t ldw, int n_mt, int k_tiles,
                                                       float* __restrict__ dst) {
    const int64_t total = (int64_t)n_mt * k_tiles * 1024;  // floats
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int c = (int)(e & 3);
        int lane = (int)((e >> 2) & 63);
        int64_t blk = e >> 8;
        int mt = (int)(blk % n_mt);
        int64_t t = blk / n_mt;
        int rho = (int)(t & 3);
        int kt = (int)(t >> 2);
        int i = lane & 31, h = lane >> 5;
        dst[e] = W[(int64_t)(32 * mt + i) * ldw + 32 * kt + 8 * rho + 4 * h + c];
    }
}

}  // namespace psn

extern "C" int psn_mlp_pack_layer(const float* W, int64_t ldw, int n_mt, int k_tiles, float* dst, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(W && dst, "mlp_pack_layer: null pointer");
    PSN_CHECK_ARG(n_mt >= 1 && n_mt <= 8 && k_tiles >= 1 && k_tiles <= 12, "mlp_pack_layer: n_mt=%d k_tiles=%d", n_mt, k_tiles);
    PSN_CHECK_ARG(ldw >= (int64_t)k_tiles * 32, "mlp_pack_layer: ldw too small");
    int64_t total = (int64_t)n_mt * k_tiles * 1024;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, ldw, n_mt, k_tiles, dst);
    PSN_CHECK_LAUNCH("mlp_pack_layer");
    return PSN_OK;
}

extern "C" int psn_mlp_infer(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                             int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                             int64_t n_rows, float* out, void* stream) {
    using namespace psn;
    PSN_CHECK_ARG(desc && packed_w && packed_b && tab_a && out, "mlp_infer: null pointer");
    const PsnMlpDesc& d = *desc;
    PSN_CHECK_ARG(d.n_layers >= 1 && d.n_layers <= PSN_MLP_MAX_LAYERS, "mlp_infer: n_layers=%d", d.n_layers);
    PSN_CHECK_ARG(d.in_kt_a >= 1 && d.in_kt_b >= 0 && d.in_kt_a + d.in_kt_b <= 4, "mlp_infer: input tiles %d+%d", d.in_kt_a, d.in_kt_b);
    PSN_CHECK_ARG(d.in_kt_b == 0 || tab_b, "mlp_infer: table B missing");
    PSN_CHECK_ARG(d.n_out >= 1 && d.n_out <= 32, "mlp_infer: n_out=%d", d.n_out);
    PSN_CHECK_ARG(a_div >= 1 && a_mod >= 1 && (d.in_kt_b == 0 || (b_div >= 1 && b_mod >= 1)), "mlp_infer: bad index map");
    PSN_CHECK_ARG((((uintptr_t)tab_a | (uintptr_t)tab_b | (uintptr_t)packed_w | (uintptr_t)packed_b) & 15) == 0,
                  "mlp_infer: buffers must be 16-byte aligned");
    for (int l = 0; l < d.n_layers; ++l) {
        const PsnMlpLayer& L = d.layers[l];
        const bool last = l == d.n_layers - 1;
        PSN_CHECK_ARG(L.n_mt == (last ? 1 : 8), "mlp_infer: layer %d n_mt=%d (hidden layers are 256 wide, final <= 32)", l, L.n_mt);
        PSN_CHECK_ARG(L.n_kt_in >= 0 && L.n_kt_in <= d.in_kt_a + d.in_kt_b, "mlp_infer: layer %d n_kt_in=%d", l, L.n_kt_in);
        PSN_CHECK_ARG(L.n_kt_act == 0 || L.n_kt_act == 8, "mlp_infer: layer %d n_kt_act=%d", l, L.n_kt_act);
        PSN_CHECK_ARG(l > 0 || L.n_kt_act == 0, "mlp_infer: layer 0 cannot read activations");
        PSN_CHECK_ARG(L.n_kt_in + L.n_kt_act >= 1, "mlp_infer: layer %d has no input", l);
        PSN_CHECK_ARG((L.w_off % 4) == 0 && (L.b_off % 4) == 0, "mlp_infer: layer %d offsets must be multiples of 4 floats", l);
    }
    if (n_rows <= 0) return PSN_OK;
    InferArgs a;
    a.d = d; a.w = packed_w; a.b = packed_b; a.ta = tab_a; a.a_div = a_div; a.a_mod = a_mod;
    a.tb = tab_b; a.b_div = b_div > 0 ? b_div : 1; a.b_mod = b_mod > 0 ? b_mod : 1; a.n_rows = n_rows; a.out = out;
    int64_t blocks = (n_rows + 127) / 128;
    PSN_CHECK_ARG(blocks < (1ll << 31), "mlp_infer: too many rows");
    hipLaunchKernelGGL(mlp_infer_kernel, dim3((unsigned)blocks), dim3(256), 2 * kStageFloats * sizeof(float),
                       (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("mlp_infer");
    return PSN_OK;
}
