"""cProfile INSIDE ops.VisibilityPair.backward (it runs on the autograd engine's thread)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops
from psnerf_amd.synthetic import stage2_inputs
pr = cProfile.Profile()
on = [False]
target = getattr(ops, sys.argv[1] if len(sys.argv) > 1 else 'VisibilityPair')
f = target.backward
def g(*a, **k):
    if on[0]: pr.enable()
    try:
        return f(*a, **k)
    finally:
        if on[0]: pr.disable()
target.backward = staticmethod(g)
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); on[0] = True
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats(30); print(s.getvalue()[:7000])
