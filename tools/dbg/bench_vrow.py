"""Ablations of the V-row backward chain (8 x 29487 rows, 8-layer ReLU-mask chain): where do the 0.4 ms over the lean rate go?"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns = 29487
Q = 8 * Ns
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
H = [torch.randn(Q, 256, device=dev) for _ in range(8)]
DZ = [torch.empty(Q, 256, device=dev) for _ in range(8)]
g = torch.randn(Q, 1, device=dev)
wl = ws[-1].contiguous()
init = hip.gemm(g, wl)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def pack(act):
    zeros = fused._zeros(256, dev)
    layers = [dict(init_a=fused.DIRECT_INIT, init_b=None, w_act=None, bias=zeros, act=act)]
    for l in range(7, 0, -1):
        layers.append(dict(w_act=fused.Transposed(ws[l][:, :256]), bias=zeros, act=act))
    return fused.pack_layers(layers, 1, 0, 0, hip.OUT_NONE, dev, has_final=False)


full = pack(hip.ACT_RELU_MASK)
relu = pack(hip.ACT_RELU)
best = {}
for rep in range(3):
    for k, fn in (
        ('mask + dump, rank init (product)', lambda: full(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), mask=H, save=DZ)),
        ('mask + dump, init table', lambda: full(None, Q, a_div=1, a_mod=Q, init_a_direct=init, mask=H, save=DZ)),
        ('mask, no dump', lambda: full(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), mask=H, save=[None] * 7 + [DZ[7]])),
        ('plain ReLU + dump (chain flavour)', lambda: relu(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), save=DZ)),
        ('plain ReLU, one dump (chain flavour)', lambda: relu(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), save=[None] * 7 + [DZ[7]])),
    ):
        best[k] = min(best.get(k, 1e9), t(fn))
fl = 2.0 * 7 * 65536 * Q
for k, v in best.items():
    print('%-40s %.3f ms  %6.1f TF' % (k, v, fl / v * 1e-9))
