"""gemm_tn256_grouped_kernel: time vs K and item count (is there a fixed per-launch cost?)."""
import sys, torch
sys.path.insert(0, '.')
from psnerf_amd import hip
dev = torch.device('cuda')
torch.manual_seed(0)
def run(K, n_items, iters=5):
    A = [torch.randn(K, 256, device=dev) for _ in range(n_items)]
    B = [torch.randn(K, 256, device=dev) for _ in range(n_items)]
    items = [dict(A=a, B=b, colsum=True) for a, b in zip(A, B)]
    for _ in range(2): hip.gemm_tn_grouped(items)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hip.gemm_tn_grouped(items)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('K=%7d items=%2d  %.3f ms  %.1f TF' % (K, n_items, ms, 2.0 * K * 65536 * n_items / ms / 1e9))
for K, n in ((235896, 6), (471792, 6), (943584, 6), (235896, 12), (524288, 16), (131072, 16), (65536, 6)):
    run(K, n)
