"""HBM calibration with plain torch kernels (copy = 1:1 read/write, sum = read only, fill = write only)."""
import torch
dev = torch.device('cuda:0')
n = 1 << 30  # 4 GB fp32
a = torch.rand(n, device=dev); b = torch.empty_like(a)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = t(lambda: b.copy_(a)); print('copy  %.3f ms  %.0f GB/s (read + write)' % (ms, 8 * n / ms * 1e-6))
ms = t(lambda: a.sum()); print('sum   %.3f ms  %.0f GB/s (read)' % (ms, 4 * n / ms * 1e-6))
ms = t(lambda: b.fill_(1.0)); print('fill  %.3f ms  %.0f GB/s (write)' % (ms, 4 * n / ms * 1e-6))
ms = t(lambda: torch.add(a, 1.0, out=b)); print('add   %.3f ms  %.0f GB/s (read + write)' % (ms, 8 * n / ms * 1e-6))
