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

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float softplus100(float z) {
    // torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20 else log1p(exp(beta*x))/beta
    float bz = z * 100.0f;
    return bz > 20.0f ? z : log1pf(expf(bz)) / 100.0f;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace psn
