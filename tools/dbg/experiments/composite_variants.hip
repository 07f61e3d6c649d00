// Microbenchmark of forward alpha-composite layouts (weights kept): the shipped kernels of csrc/composite.hip against
// multi-ray-per-wave variants.  2 M rays x S samples by default (working set >> Infinity Cache).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/dbg/experiments/composite_variants.hip psnerf_amd/csrc/error.hip -o tools/dbg/bin/composite_variants
//   tools/dbg/bin/composite_variants [rays] [S]
#include "../../../psnerf_amd/csrc/composite.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

namespace psn {

// ceiling probes: the same bytes as the forward (4 S read as alpha + 12 S as colours, 4 S written) moved by a plain streaming
// kernel, float4 per lane, grid-stride: RD read streams of equal length summed into WR write streams
template <int RD, int WR, bool NT>
__global__ __launch_bounds__(256) void stream_rw_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const float4 v = NT ? nt_load4(reinterpret_cast<const float*>(in + r * n4 + i)) : in[r * n4 + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
#pragma unroll
        for (int q = 0; q < WR; ++q) {
            if (NT) nt_store4(reinterpret_cast<float*>(out + q * n4 + i), acc.x + q, acc.y, acc.z, acc.w);
            else out[q * n4 + i] = make_float4(acc.x + q, acc.y, acc.z, acc.w);
        }
    }
}

}  // namespace psn

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    using namespace psn;
    const int64_t N = argc > 1 ? atoll(argv[1]) : 2 * 1024 * 1024;
    const int S = argc > 2 ? atoi(argv[2]) : 128;
    float *alpha, *rgb, *w, *w_ref, *ro, *ro_ref, *ao, *ao_ref;
    CK(hipMalloc(&alpha, N * S * 4)); CK(hipMalloc(&rgb, N * S * 12)); CK(hipMalloc(&w, N * S * 4)); CK(hipMalloc(&w_ref, N * S * 4));
    CK(hipMalloc(&ro, N * 12)); CK(hipMalloc(&ro_ref, N * 12)); CK(hipMalloc(&ao, N * 4)); CK(hipMalloc(&ao_ref, N * 4));
    {
        std::vector<float> h((size_t)N * S * 3);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
        for (size_t i = 0; i < (size_t)N * S; ++i) h[i] = rnd() * 0.1f;
        CK(hipMemcpy(alpha, h.data(), N * S * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)N * S * 3; ++i) h[i] = rnd();
        CK(hipMemcpy(rgb, h.data(), N * S * 12, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (20.0 * S + 16.0) * N;
    auto run = [&](const char* name, auto launch, bool is_ref) {
        CK(hipMemset(w, 0xff, N * S * 4)); CK(hipMemset(ro, 0xff, N * 12)); CK(hipMemset(ao, 0xff, N * 4));
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0.f;
        const int it = 10;
        for (int k = 0; k < it; ++k) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        double maxd = -1.0;
        if (is_ref) {
            CK(hipMemcpy(w_ref, w, N * S * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ro_ref, ro, N * 12, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ao_ref, ao, N * 4, hipMemcpyDeviceToDevice));
        } else {
            // compare a sample of rays on the host
            const int64_t n_chk = N < 4096 ? N : 4096;
            std::vector<float> a(n_chk * S), b(n_chk * S), c(n_chk * 3), d(n_chk * 3), e(n_chk), f(n_chk);
            const int64_t off = N - n_chk;
            CK(hipMemcpy(a.data(), w + off * S, n_chk * S * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), w_ref + off * S, n_chk * S * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c.data(), ro + off * 3, n_chk * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), ro_ref + off * 3, n_chk * 12, hipMemcpyDeviceToHost));
            CK(hipMemcpy(e.data(), ao + off, n_chk * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f.data(), ao_ref + off, n_chk * 4, hipMemcpyDeviceToHost));
            maxd = 0.0;
            for (size_t i = 0; i < a.size(); ++i) { double x = std::fabs((double)a[i] - b[i]); if (!(x <= maxd)) maxd = x; }
            for (size_t i = 0; i < c.size(); ++i) { double x = std::fabs((double)c[i] - d[i]); if (!(x <= maxd)) maxd = x; }
            for (size_t i = 0; i < e.size(); ++i) { double x = std::fabs((double)e[i] - f[i]); if (!(x <= maxd)) maxd = x; }
        }
        printf("{\"variant\": \"%s\", \"avg_ms\": %.4f, \"min_ms\": %.4f, \"avg_TBps\": %.3f, \"max_TBps\": %.3f, \"max_abs_diff_vs_shipped\": %.3g}\n", name, sum / it, best,
               bytes / (sum / it) * 1e-9, bytes / best * 1e-9, maxd);
    };
    run("shipped (psn_composite_fwd)", [&]() { psn_composite_fwd(alpha, rgb, N, S, 1, w, ro, ao, nullptr); }, true);
    auto grid_for = [&](int rpw, int mult) { int64_t b = (N + 4 * rpw - 1) / (4 * rpw); int64_t cap = 256 * mult; return (unsigned)(b > cap ? cap : b); };
#define RUNV(NAME, LPR, MODE, DEPTH, MULT) \
    run(NAME, [&]() { hipLaunchKernelGGL((composite_fwd_multi_kernel<LPR, MODE, DEPTH>), dim3(grid_for(64 / LPR, MULT)), dim3(256), 0, 0, alpha, rgb, N, S, 1, w, ro, ao); }, false);
    if (S <= 128) {
        RUNV("multi LPR32 temporal depth1 grid x8", 32, 0, 1, 8)
        RUNV("multi LPR32 temporal depth1 grid x16", 32, 0, 1, 16)
        RUNV("multi LPR32 nt depth1 grid x8", 32, 1, 1, 8)
        RUNV("multi LPR32 nt depth1 grid x16", 32, 1, 1, 16)
        RUNV("multi LPR32 temporal depth2 grid x8", 32, 0, 2, 8)
        RUNV("multi LPR32 nt depth2 grid x8", 32, 1, 2, 8)
        RUNV("multi LPR32 nt depth2 grid x16", 32, 1, 2, 16)
    }
    if (S <= 64) {
        RUNV("multi LPR16 temporal depth1 grid x8", 16, 0, 1, 8)
        RUNV("multi LPR16 nt depth1 grid x8", 16, 1, 1, 8)
        RUNV("multi LPR16 nt depth2 grid x8", 16, 1, 2, 8)
    }
    {
        // alpha and rgb are separate allocations: copy them into one 4-stream buffer of equal streams (alpha | rgb thirds)
        const int64_t n4 = N * S / 4;
        float4* in4; float4* out4;
        CK(hipMalloc(&in4, n4 * 16 * 4)); CK(hipMalloc(&out4, n4 * 16 * 2));
        CK(hipMemcpy(in4, alpha, n4 * 16, hipMemcpyDeviceToDevice)); CK(hipMemcpy(in4 + n4, rgb, n4 * 16 * 3, hipMemcpyDeviceToDevice));
        auto probe = [&](const char* name, auto launch, double by) {
            launch(); CK(hipDeviceSynchronize());
            float sum = 0.f, best = 1e9f;
            for (int k = 0; k < 10; ++k) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = ms < best ? ms : best; }
            printf("{\"probe\": \"%s\", \"avg_ms\": %.4f, \"avg_TBps\": %.3f, \"max_TBps\": %.3f}\n", name, sum / 10, by / (sum / 10) * 1e-9, by / best * 1e-9);
        };
        const double b41 = 20.0 * S * N, b40 = 16.0 * S * N, b11 = 8.0 * S * N, b21 = 12.0 * S * N;
        probe("stream 4 reads + 1 write, temporal, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<4, 1, false>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b41);
        probe("stream 4 reads + 1 write, temporal, grid 256x32", [&]() { hipLaunchKernelGGL((stream_rw_kernel<4, 1, false>), dim3(8192), dim3(256), 0, 0, in4, out4, n4); }, b41);
        probe("stream 4 reads + 1 write, nt, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<4, 1, true>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b41);
        probe("stream 4 reads + 1 write, nt, grid 256x32", [&]() { hipLaunchKernelGGL((stream_rw_kernel<4, 1, true>), dim3(8192), dim3(256), 0, 0, in4, out4, n4); }, b41);
        probe("stream 4 reads + 0 writes, temporal, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<4, 0, false>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b40);
        probe("stream 1 read + 1 write, temporal, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<1, 1, false>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b11);
        probe("stream 2 reads + 1 write, temporal, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<2, 1, false>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b21);
        probe("stream 2 reads + 1 write, nt, grid 256x8", [&]() { hipLaunchKernelGGL((stream_rw_kernel<2, 1, true>), dim3(2048), dim3(256), 0, 0, in4, out4, n4); }, b21);
    }
    run("shipped again", [&]() { psn_composite_fwd(alpha, rgb, N, S, 1, w, ro, ao, nullptr); }, false);
    return 0;
}
