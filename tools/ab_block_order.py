"""A/B of the visibility launch's workgroup order (psn_mlp_block_order): the stage-2 shading + supervision rows at the bench size
(29,487 surface points x (96 + 8) lights, dumps of the 8 supervised groups), fp32 and split-bf16 weight stages, row order vs
point-tile-major.  Kernel times by HIP events; outputs compared bit for bit.   python tools/ab_block_order.py [--order row|point]
(--order: one order only, for a rocprofv3 --pmc FETCH_SIZE pass per order)"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psnerf_amd import fused, hip

ap = argparse.ArgumentParser()
ap.add_argument('--order', default=None)
ap.add_argument('--ns', type=int, default=29487)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--no-x3', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
Ns, L, V = a.ns, 96, 8
dims = [(256, 78)] + [(256, 256)] * 3 + [(256, 256 + 78)] + [(256, 256)] * 2 + [(1, 256)]
Ws = [torch.randn(o, i, device=dev) * (1.4 / i ** 0.5) for o, i in dims]
bs = [torch.randn(o, device=dev) * 0.01 for o, _ in dims]
pe_x, pe_l = torch.randn(Ns, 64, device=dev), torch.randn(L + V, 64, device=dev)


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rep = {'Ns': Ns, 'L': L, 'V': V}
for x3 in ((False,) if a.no_x3 else (False, True)):
    pk = fused.pack_relu_mlp(Ws, bs, 39, 39, 3, x3=x3)
    save = [torch.empty(V * Ns, 256, device=dev) for _ in range(len(Ws) - 1)]
    outs = {}
    for order in ([a.order] if a.order else ['row', 'point', 'row', 'point']):
        out = torch.full(((L + V) * Ns, 1), float('nan'), device=dev)
        f = lambda: pk(pe_x, (L + V) * Ns, a_div=1, a_mod=Ns, tab_b=pe_l, b_div=Ns, b_mod=L + V, out=out, save=save, save_row0=L * Ns)
        with hip.block_order(order):
            ms = timeit(f, a.iters)
        key = ('bf16x3' if x3 else 'fp32') + '_' + order + '_ms'
        rep.setdefault(key, []).append(round(ms, 3))
        outs[order] = (out.clone(), save[3].clone())
    if len(outs) == 2:
        rep[('bf16x3' if x3 else 'fp32') + '_bit_identical'] = bool(torch.equal(outs['row'][0], outs['point'][0]) and torch.equal(outs['row'][1], outs['point'][1]))
print(json.dumps(rep))
