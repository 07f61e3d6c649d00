// Shared helpers for libpsnerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/psnerf_hip.h"

namespace psn {

void set_error(const char* fmt, ...);

#define PSN_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            psn::set_error(__VA_ARGS__);  \
            return PSN_E_ARG;             \
        }                                 \
    } while (0)

#define PSN_CHECK_LAUNCH(what)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            psn::set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return PSN_E_LAUNCH;                                                 \
        }                                                                        \
    } while (0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + barrier and its workgroup-scope release is
// lowered to `s_waitcnt vmcnt(0) lgkmcnt(0)`: every barrier then also waits for all of the wave's outstanding GLOBAL
// loads and stores -- which defeats any prefetch that is meant to stay in flight across a k-tile (gemm_tn256: the rows
// of tile t + 3) or dump stores that should drain under the next layer (chain engine).  Use this when the barrier only
// publishes LDS writes / retires LDS reads; data hand-offs through global memory still need __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// LDS-DMA of one 1 KB piece (64 lanes x 16 B): global (gbase + voff + OFF) -> LDS (lds + OFF + 16 * lane); gbase and lds
// wave-uniform (SGPRs), voff = 16 * lane.  Inline assembly on purpose: hipcc models the builtin
// (__builtin_amdgcn_global_load_lds, a FLAT-encoded instruction with a global AND an LDS memory operand) as a "flat access
// that may touch both address spaces", and while one is pending its waitcnt pass turns EVERY lgkmcnt wait of the wave into
// lgkmcnt(0) -- the fragment prefetches of the fused MLP engines, meant to stay in flight for hundreds of cycles, were
// drained at each use (one exposed LDS latency per 4-12 MFMAs).  An asm statement is invisible to that pass: the DS
// waits become exact, and the landing of the pieces is waited for explicitly (s_waitcnt vmcnt) before the stage barrier
// as before.  The scalar-base form also reads one address VGPR instead of two.
template <int OFF>
__device__ __forceinline__ void lds_dma_16(const void* gbase, unsigned lds, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(gbase), "s"(lds), "n"(OFF));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const void*)p;
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// max(x, 0) in ONE instruction.  fmaxf(x, 0.0f) on an MFMA result compiles to two (v_max_f32 x, x, x to quiet a
// possible signalling NaN, then the max; v_med3 / maxnum intrinsics are folded back to the same pair): in the fp32
// MFMA kernels every vector instruction costs matrix-pipe time, and ReLU runs on every activation.  v_max_f32 itself
// already returns the non-NaN operand, i.e. 0 for a NaN input, like fmaxf.
__device__ __forceinline__ float relu1(float x) {
    float o;
    asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(x));
    return o;
}

// exp(-|t|) with a Cody-Waite split of the exp2 argument: v_exp_f32 is ~1 ulp on 2^x, the split keeps
// the product |t|*log2(e) exact to ~2^-48, so u is accurate to ~2 ulp for |t| <= 100.
__device__ __forceinline__ float exp_neg_abs(float at) {
    const float L2E_HI = 1.44269502162933349609375f;    // (float) log2(e)
    const float L2E_LO = 1.925963033500011e-8f;          // log2(e) - L2E_HI
    float hi = -at * L2E_HI;
    float lo = fmaf(-at, L2E_HI, -hi);
    lo = fmaf(-at, L2E_LO, lo);
    float u = __builtin_amdgcn_exp2f(hi);
    return fmaf(u, lo * 0.693147182464599609375f, u);
}

// log1p(u) for u in [0, 1] from the hardware log2: log(w) with w = fl(1 + u), plus the classic rounding correction
// (u - (w - 1)) / w for what the addition lost.  Measured on gfx950 over u = exp(-|t|), |t| <= 40 (tools/dbg/
// check_log1p.py): max relative error 1.5e-7 on every sub-range down to u ~ 1e-17, so no separate small-u series is
// needed (the earlier one cost 7 FMAs + a select per element; every vector instruction costs matrix-pipe time in the
// fp32 MFMA kernels).  The correction term is O(2^-24) of the result: a 1-ulp reciprocal is plenty.
__device__ __forceinline__ float log1p_unit(float u, float w, float rw) {  // w = 1 + u, rw ~ 1 / w (shared with the sigmoid)
    return fmaf(__builtin_amdgcn_logf(w), 0.693147182464599609375f, (u - (w - 1.0f)) * rw);
}

// torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20 else log1p(exp(beta*x))/beta, evaluated as
// max(x,0) + log1p(exp(-|beta x|))/beta (identical in exact arithmetic, no overflow, ~25 instructions).
// Also returns s = sigmoid(beta*x) = d softplus / dx from the same exponential.
__device__ __forceinline__ void softplus100_sig(float z, float& sp, float& s) {
    float t = z * 100.0f;
    float u = exp_neg_abs(fabsf(t));
    const float w = 1.0f + u;
    const float rw = __builtin_amdgcn_rcpf(w);     // 1 ulp
    float l = log1p_unit(u, w, rw);
    float r = fmaf(fmaf(-w, rw, 1.0f), rw, rw);    // one Newton step: 1 / w to ~0.5 ulp without the IEEE division sequence
    s = t >= 0.0f ? r : u * r;
    // x / 100 correctly rounded without the IEEE division sequence (its length makes hipcc branch around it)
    float x = relu1(t) + l;
    float q = x * 0.01f;
    q = fmaf(fmaf(-q, 100.0f, x), 0.01f, q);
    sp = t > 20.0f ? z : q;
}
__device__ __forceinline__ float softplus100(float z) {
    float sp, s;
    softplus100_sig(z, sp, s);
    return sp;
}
__device__ __forceinline__ float sigmoidf_(float x) {
    float u = exp_neg_abs(fabsf(x));
    float r = 1.0f / (1.0f + u);
    return x >= 0.0f ? r : u * r;
}

}  // namespace psn
