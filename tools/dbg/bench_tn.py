import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip
dev = torch.device('cuda:0')
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
a = torch.randn(Q, 256, device=dev); b = torch.randn(Q, 256, device=dev)
cs = torch.empty(256, device=dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for sk in (32, 64, 128, 256, 512):
    ms = t(lambda: hip.gemm(a, b, trans_a=True, split_k=sk, colsum_a=cs))
    print('BN=%s split %4d: %.3f ms  %.1f TF' % (os.environ.get('PSN_GEMM_BN', 'auto'), sk, ms, 2 * 256 * 256 * Q / ms / 1e9))

items = [dict(A=torch.randn(Q, 256, device=dev), B=torch.randn(Q, 256, device=dev), A2=torch.randn(Q, 256, device=dev), B2=torch.randn(Q, 256, device=dev), colsum=True) for _ in range(8)]
for sk in (2, 4, 8, 16, 32):
    ms = t(lambda: hip.gemm_tn_grouped(items, sk))
    print('grouped 8 x 2 products, split %3d: %.3f ms  %.1f TF' % (sk, ms, 16 * 2 * 256 * 256 * Q / ms / 1e9))
