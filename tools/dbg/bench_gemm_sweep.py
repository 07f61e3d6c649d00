import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip
dev = torch.device('cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, K) in ((236000, 256), (59000, 1024), (14750, 4096), (236000, 64), (236000, 128)):
    A = torch.randn(M, K, device=dev); W = torch.randn(256, K, device=dev); C = torch.empty(M, 256, device=dev)
    ms = t(lambda: hip.gemm(A, W, out=C, trans_b=True))
    print('NT M=%d K=%d N=256: %.3f ms %.1f TF' % (M, K, ms, 2 * M * K * 256 / ms / 1e9))
M = 236000
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); C = torch.empty(M, 256, device=dev)
ms = t(lambda: C.copy_(A)); print('copy 242MB: %.3f ms %.0f GB/s' % (ms, 2 * M * 1024 / ms / 1e6))
ms = t(lambda: torch.relu(A)); print('relu 242MB: %.3f ms' % ms)
