// Microbenchmark: HBM throughput of the chain engine's dump / operand access pattern.
//   pattern 0: row-major [rows, 256] fp32, lane (i = lane & 15, g = lane >> 4) touches row 16*tile + i, floats 16*mt + 4*g .. +3
//              (what mlp_infer_kernel<true> does today: 16 segments of 64 B per wave-instruction, 1 KB apart)
//   pattern 1: tile-major [rows / 16][16 mt][64 lanes][4] fp32: every wave-instruction moves one contiguous 1 KB
// Each wave streams NR operand tensors in and NW dump tensors out for its 16 rows x 8 "layers".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PATTERN, int NR, int NW>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, long rows, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 4 + wave;  // 16 rows
    if (tile * 16 >= rows) return;
    const int i = lane & 15, g = lane >> 4;
    const long per_tensor = rows * 256;
    for (int l = 0; l < layers; ++l) {
        float4 v[NR > 0 ? NR : 1][16];
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const float* base = in + (long)(l * NR + t) * per_tensor;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const float* p = PATTERN == 0 ? base + (tile * 16 + i) * 256 + mt * 16 + 4 * g : base + tile * 4096 + mt * 256 + lane * 4;
                v[t][mt] = *reinterpret_cast<const float4*>(p);
            }
        }
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            float* base = out + (long)(l * NW + t) * per_tensor;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                float4 o = NR > 0 ? v[t % (NR > 0 ? NR : 1)][mt] : make_float4(l, t, mt, lane);
                if (NR > 1) { o.x += v[(t + 1) % NR][mt].x; o.y *= v[(t + 1) % NR][mt].y; }
                float* p = PATTERN == 0 ? base + (tile * 16 + i) * 256 + mt * 16 + 4 * g : base + tile * 4096 + mt * 256 + lane * 4;
                *reinterpret_cast<float4*>(p) = o;
            }
        }
    }
}

template <int PATTERN, int NR, int NW>
void run(const char* name, float* in, float* out, long rows, int layers) {
    dim3 grid((unsigned)((rows / 16 + 3) / 4)), block(256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<PATTERN, NR, NW>), grid, block, 0, 0, in, out, rows, layers);
    hipEventRecord(a);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<PATTERN, NR, NW>), grid, block, 0, 0, in, out, rows, layers);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    double bytes = (double)rows * 1024.0 * layers * (NR + NW);
    printf("%-28s pattern %d  %d in %d out: %8.3f ms  %7.1f GB/s\n", name, PATTERN, NR, NW, ms, bytes / ms * 1e-6);
}

int main(int argc, char** argv) {
    long rows = argc > 1 ? atol(argv[1]) : 262144;
    int layers = 8;
    size_t n = (size_t)rows * 256 * layers * 2;
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4); hipMemset(out, 0, n * 4);
    printf("rows %ld, %d layers, %.1f GB per tensor set\n", rows, layers, n * 4 / 1e9);
    run<0, 2, 2>("B1-like row-major", in, out, rows, layers);
    run<1, 2, 2>("B1-like tile-major", in, out, rows, layers);
    run<0, 1, 2>("F2-like row-major", in, out, rows, layers);
    run<1, 1, 2>("F2-like tile-major", in, out, rows, layers);
    run<0, 0, 2>("F1-like row-major", in, out, rows, layers);
    run<1, 0, 2>("F1-like tile-major", in, out, rows, layers);
    run<0, 2, 0>("read-only row-major", in, out, rows, layers);
    run<1, 2, 0>("read-only tile-major", in, out, rows, layers);
    return 0;
}
