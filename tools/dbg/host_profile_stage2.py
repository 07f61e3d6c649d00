"""cProfile of the HOST side of a stage-2 train step (tiny batch: the GPU is never the bottleneck) -> top functions by own / cumulative time."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5):
    step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100):
    step.step(inp, gt, l_slt, train_order=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host issue time per step: %.3f ms' % ((t1 - t0) * 10))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step.step(inp, gt, l_slt, train_order=False)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
