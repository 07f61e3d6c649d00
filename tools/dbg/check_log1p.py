"""Accuracy of the two log1p formulas of csrc/common.h (hardware log2 + correction vs short series) over u in (0, 1]."""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'liblg.so'))
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
