"""Eager stage-2 step on a resident batch vs four resident batches with DIFFERENT surface counts cycled (what a sampler hands out):
is the eager step itself sensitive to the changing shapes (allocator / workspace caches), apart from any sampler cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
l_slt = torch.arange(96, device=dev) + 288
bs = [stage2_inputs(32768, 96, 8, seed=100 + i, device=dev, with_surface_idx=True) for i in range(4)]
print('surface counts', [int(b[0]['surface_idx'].numel()) for b in bs])
def run(batches, n=40):
    for i in range(8):
        step.step(*batches[i % len(batches)], l_slt, train_order=False)
    bench.settle_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step.step(*batches[i % len(batches)], l_slt, train_order=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3
print('same batch      ms/step %.3f (host issue %.3f)' % run(bs[:1]))
print('4 batches cycled ms/step %.3f (host issue %.3f)' % run(bs))
print('same batch      ms/step %.3f (host issue %.3f)' % run(bs[:1]))
