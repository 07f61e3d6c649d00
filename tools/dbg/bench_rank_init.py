"""Backward chains started from a materialised init table vs the rank-k init formed inside the kernel."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Q = 524288
Ws = [torch.randn(256, 289, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(3, 256, device=dev) * 0.06]
chain = fused.pack_relu_bwd(Ws, -100)
H = [torch.randn(Q, 256, device=dev) for _ in range(4)]
DZ = [torch.empty(Q, 256, device=dev) for _ in range(4)]
g = torch.randn(Q, 3, device=dev)
Wl = Ws[-1].contiguous()


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def table():
    dh = hip.gemm(g, Wl)
    chain(None, Q, a_div=1, a_mod=Q, init_a_direct=dh, mask=H, save=DZ)


def rank():
    chain(None, Q, a_div=1, a_mod=Q, rank_init=(g, Wl), mask=H, save=DZ)


for rep in range(3):
    table(); ref = [d.clone() for d in DZ]
    rank()
    err = max(float((a - b).abs().max()) for a, b in zip(DZ, ref))
    print('ReLU backward chain, 4 layers, %d rows: init table (K = 3 GEMM + chain) %.3f ms, rank-3 init %.3f ms, max|d| %.2e' % (Q, t(table), t(rank), err))
