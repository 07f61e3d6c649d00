"""Accuracy of the two log1p formulas of csrc/common.h (hardware log2 + correction vs short series) over u in (0, 1]."""
import ctypes, os, subprocess, sys, tempfile
import torch
SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(const float* u_in, float* big_out, float* ser_out, float* rcp_out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u = u_in[i];
    float w = 1.0f + u;
    float rw = __builtin_amdgcn_rcpf(w);
    big_out[i] = fmaf(__builtin_amdgcn_logf(w), 0.693147182464599609375f, (u - (w - 1.0f)) * rw);
    float p = fmaf(-u, 0.142857149f, 0.166666672f);
    p = fmaf(-u, p, 0.2f); p = fmaf(-u, p, 0.25f); p = fmaf(-u, p, 0.333333343f); p = fmaf(-u, p, 0.5f); p = fmaf(-u, p, 1.0f);
    ser_out[i] = u * p;
    rcp_out[i] = fmaf(fmaf(-w, rw, 1.0f), rw, rw);
}
extern "C" int run(const float* u, float* b, float* s, float* r, int n, void* st) {
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, u, b, s, r, n);
    return (int)hipGetLastError();
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, 'lg.hip'), 'w').write(SRC)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-shared',
                       os.path.join(tmp, 'lg.hip'), '-o', os.path.join(tmp, 'liblg.so')])
lib = ctypes.CDLL(os.path.join(tmp, 'liblg.so'))
lib.run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda:0')
n = 1 << 22
u = torch.exp(-torch.rand(n, device=dev, dtype=torch.float64) * 40).float()  # e^-40 .. 1, log-uniform (= exp(-|t|), |t| <= 40)
b, s, r = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
assert lib.run(u.data_ptr(), b.data_ptr(), s.data_ptr(), r.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
ref = torch.log1p(u.double())
for lo, hi in ((0, 1e-6), (1e-6, 1e-3), (1e-3, 0.0625), (0.0625, 0.25), (0.25, 1.0)):
    m = (u > lo) & (u <= hi)
    eb = ((b.double() - ref).abs() / ref)[m].max().item()
    es = ((s.double() - ref).abs() / ref)[m].max().item()
    print('u in (%g, %g]: %7d samples   log2 + correction: max rel err %.2e   series: %.2e' % (lo, hi, int(m.sum()), eb, es))
rr = 1.0 / (1.0 + u.double())
print('Newton reciprocal of 1 + u: max rel err %.2e' % ((r.double() - rr).abs() / rr).max().item())
