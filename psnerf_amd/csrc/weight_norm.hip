// Weight normalisation of all layers of a network in one launch: the effective matrices of
// nn.utils.weight_norm(nn.Linear) -- stage1/model/network.py:37-66 wraps every layer of the geometry and appearance
// networks in it -- and their backward.  w = v * (g / |v|_row) [* scale], one wave per weight row.
// The torch formulation costs ~4 small kernels per layer forward and ~12 backward (14 layers, evaluated 3-4 times per
// train step: ~460 launches, 2.3 ms of a 57 ms step); this is 1 + 1 per network.  HBM-bound on ~3 MB of weights.
#include "common.h"

namespace psn {

struct WnArgs {
    PsnWnItem it[PSN_WN_MAX_ITEMS];
    int row_start[PSN_WN_MAX_ITEMS + 1];
    int n;
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

template <bool BWD>
__global__ __launch_bounds__(256) void weight_norm_kernel(WnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row_g >= a.row_start[a.n]) return;
    int i = 0;
    while (i + 1 < a.n && row_g >= a.row_start[i + 1]) ++i;
    const PsnWnItem it = a.it[i];
    const int r = row_g - a.row_start[i];
    const float* v = it.v + (int64_t)r * it.cols;
    float n2 = 0.f;
    for (int c = lane; c < it.cols; c += 64) n2 += v[c] * v[c];
    const float nrm = sqrtf(wave_sum(n2));
    const float g = it.g[r];
    const float s = g / nrm;
    if (!BWD) {
        float* w = it.w + (int64_t)r * it.cols;
        if (it.scale == 1.0f) {
            for (int c = lane; c < it.cols; c += 64) w[c] = v[c] * s;
        } else {
            for (int c = lane; c < it.cols; c += 64) w[c] = (v[c] * s) * it.scale;  // the reference's op order: v * (g / |v|), then the fold
        }
    } else {
        const float* dw = it.dw + (int64_t)r * it.cols;
        float dot = 0.f;
        for (int c = lane; c < it.cols; c += 64) dot += dw[c] * v[c];
        dot = wave_sum(dot) * it.scale;              // d s = sum_j dW_j v_j  (dW taken w.r.t. the scaled output)
        const float k = dot * g / (nrm * nrm * nrm); // d |v| = -d s * g / |v|^2 ;  d|v| / d v_j = v_j / |v|
        float* dv = it.dv + (int64_t)r * it.cols;
        const float ss = s * it.scale;
        for (int c = lane; c < it.cols; c += 64) dv[c] = dw[c] * ss - k * v[c];
        if (lane == 0) it.dg[r] = dot / nrm;
    }
}

static int launch_wn(int n_items, const PsnWnItem* items, bool bwd, void* stream) {
    PSN_CHECK_ARG(items && n_items >= 1 && n_items <= PSN_WN_MAX_ITEMS, "weight_norm: n_items=%d", n_items);
    WnArgs a;
    a.n = n_items;
    a.row_start[0] = 0;
    for (int i = 0; i < n_items; ++i) {
        const PsnWnItem& it = items[i];
        PSN_CHECK_ARG(it.v && it.g && it.rows >= 1 && it.cols >= 1, "weight_norm: item %d: null pointer or empty matrix", i);
        PSN_CHECK_ARG(bwd ? (it.dw && it.dv && it.dg) : (it.w != nullptr), "weight_norm: item %d: missing %s pointers", i, bwd ? "gradient" : "output");
        a.it[i] = it;
        a.row_start[i + 1] = a.row_start[i] + it.rows;
    }
    const int blocks = (a.row_start[n_items] + 3) / 4;
    if (bwd) hipLaunchKernelGGL(weight_norm_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(weight_norm_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    PSN_CHECK_LAUNCH("weight_norm");
    return PSN_OK;
}

}  // namespace psn

extern "C" int psn_weight_norm_fwd(int n_items, const PsnWnItem* items, void* stream) { return psn::launch_wn(n_items, items, false, stream); }
extern "C" int psn_weight_norm_bwd(int n_items, const PsnWnItem* items, void* stream) { return psn::launch_wn(n_items, items, true, stream); }
