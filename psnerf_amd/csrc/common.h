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

// ---- packed fp32 pairs -----------------------------------------------------------------------------------------------------
// In the fp32 MFMA kernels every vector instruction costs matrix-pipe time (the softplus of the stage-1 geometry network was
// ~16 % of the ray-march sweep).  gfx950 executes v_pk_{mul,add,fma}_f32 on TWO values per lane at the rate of the scalar forms,
// so the element-wise math below works on pairs; only the transcendentals, min / max and selects remain one value at a time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk2(float a) { return f32x2{a, a}; }

// torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20 else log1p(exp(beta*x))/beta, evaluated for a PAIR as
//   max(x, 0) + log1p(exp(-|beta x|)) / beta
// (identical in exact arithmetic; no overflow; no select for the threshold: above it the second term is < 2e-11 and vanishes
// in the addition, so the result is x exactly, as in torch).  Per pair: 2 x (exp2, log2) + 2 x (min, max) + 15 packed.
//   u = exp(-|t|):  hi = fl(-|t| log2 e), u = 2^hi (1 + r), r = -|t| - hi ln2 formed with a split ln2 (two FMAs: exact to
//       ~2^-48 |t|), so u carries the ~1 ulp of v_exp_f32 for |t| <= 100;
//   log1p(u) = log2(w) ln2 + (u - (w - 1)) / w,  w = fl(1 + u): the classic rounding correction; it is O(2^-24) of the result
//       unless u is tiny, where w -> 1: 1 / w ~ 1 - c + c^2 / 2 (c = w - 1; exact at c = 0 and c = 1, 6 % off in between,
//       i.e. < 1e-8 of the result) -- no reciprocal (tests/test_kernels_gpu.py::test_softplus100_accuracy_against_float64);
//   / beta: l * 0.01 with the constant split into two floats (0.01f is 2.2e-8 off 1 / 100).
// WITH_SIG: also s = sigmoid(beta x) = d softplus / dx from the same exponential (one reciprocal + a Newton step: 0.5 ulp).
template <bool WITH_SIG>
__device__ __forceinline__ void softplus100_pair(f32x2 z, f32x2& sp, f32x2& s) {
    const float L2E = 1.44269502162933349609375f;             // (float) log2(e)
    const float LN2_HI = 0.693147182464599609375f;            // (float) ln 2
    const float LN2_LO = -1.904654323148236e-9f;              // ln 2 - LN2_HI
    const float C_HI = 0.00999999977648258209228515625f;      // (float) 0.01
    const float C_LO = 2.2351741811588166e-10f;               // 0.01 - C_HI
    const f32x2 t = z * 100.0f;
    const f32x2 nat = {fminf(t.x, -t.x), fminf(t.y, -t.y)};  // -|t|
    const f32x2 hi = nat * L2E;
    f32x2 r = pk_fma(-hi, pk2(LN2_HI), nat);
    r = pk_fma(-hi, pk2(LN2_LO), r);
    const f32x2 u0 = {__builtin_amdgcn_exp2f(hi.x), __builtin_amdgcn_exp2f(hi.y)};
    const f32x2 u = pk_fma(u0, r, u0);
    const f32x2 w = u + 1.0f;
    const f32x2 lw = {__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)};  // log2
    const f32x2 c = w - 1.0f;
    const f32x2 d = u - c;
    f32x2 rw;
    if constexpr (WITH_SIG) {
        const f32x2 r0 = {__builtin_amdgcn_rcpf(w.x), __builtin_amdgcn_rcpf(w.y)};   // 1 ulp
        rw = pk_fma(pk_fma(-w, r0, pk2(1.0f)), r0, r0);                              // one Newton step: 1 / w to ~0.5 ulp
        const f32x2 ur = u * rw;
        s = f32x2{t.x >= 0.0f ? rw.x : ur.x, t.y >= 0.0f ? rw.y : ur.y};
    } else {
        rw = pk_fma(c, pk_fma(c, pk2(0.5f), pk2(-1.0f)), pk2(1.0f));
    }
    const f32x2 l = pk_fma(lw, pk2(LN2_HI), d * rw);
    const f32x2 zr = {relu1(z.x), relu1(z.y)};
    sp = pk_fma(l, pk2(C_HI), pk_fma(l, pk2(C_LO), zr));
}
// (single-value forms for the few call sites that are not in the per-layer loops)
__device__ __forceinline__ void softplus100_sig(float z, float& sp, float& s) {
    f32x2 a, b;
    softplus100_pair<true>(f32x2{z, z}, a, b);
    sp = a.x; s = b.x;
}
__device__ __forceinline__ float softplus100(float z) {
    f32x2 a, b;
    softplus100_pair<false>(f32x2{z, z}, a, b);
    return a.x;
}
__device__ __forceinline__ float sigmoidf_(float x) {
    float u = exp_neg_abs(fabsf(x));
    float r = 1.0f / (1.0f + u);
    return x >= 0.0f ? r : u * r;
}

}  // namespace psn
