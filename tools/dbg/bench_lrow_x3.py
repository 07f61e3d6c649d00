"""The stage-2 shading-row launch (visibility_net over L x Ns rows, gradient-free) on fp32 and on split-bf16 weight stages, and the
stage-1 march sweep the same way: kernel times only.   python tools/dbg/bench_lrow_x3.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import fused, hip
dev = torch.device('cuda:0')
torch.manual_seed(0)
Ns, L = 29487, 96
dims = [(256, 78)] + [(256, 256)] * 3 + [(256, 256 + 78)] + [(256, 256)] * 2 + [(1, 256)]
Ws = [torch.randn(o, i, device=dev) * (1.4 / i ** 0.5) for o, i in dims]
bs = [torch.randn(o, device=dev) * 0.01 for o, _ in dims]
pe_x, pe_l = torch.randn(Ns, 64, device=dev), torch.randn(L, 64, device=dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


rep = {}
outs = {}
for x3 in (False, True):
    pk = fused.pack_relu_mlp(Ws, bs, 39, 39, 3, x3=x3)
    out = torch.empty(L * Ns, 1, device=dev)
    f = lambda: pk(pe_x, L * Ns, a_div=1, a_mod=Ns, tab_b=pe_l, b_div=Ns, b_mod=L, out=out)
    rep['lrow_' + ('bf16x3' if x3 else 'fp32') + '_ms'] = round(timeit(f), 3)
    outs[x3] = out.clone()
rep['lrow_max_abs_diff'] = float((outs[True] - outs[False]).abs().max())
rep['lrow_out_scale'] = float(outs[False].abs().max())
print(json.dumps(rep))
