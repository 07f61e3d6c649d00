"""cProfile of ops.VisibilityPair.backward on the autograd thread (builtins included), 100 eager steps at 1024 px."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops, hip
from psnerf_amd.synthetic import stage2_inputs
pr = cProfile.Profile()
f = ops.VisibilityPair.backward
on = [False]; tot = [0.0]
def b(*a, **k):
    if not on[0]:
        return f(*a, **k)
    t = time.perf_counter()
    try: return pr.runcall(f, *a, **k)
    finally: tot[0] += time.perf_counter() - t
ops.VisibilityPair.backward = staticmethod(b)
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); on[0] = True
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
print('VisibilityPair.backward %.1f us/step (under cProfile)' % (tot[0] * 1e4))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[-3200:])
