"""Stage-2 step at batches beyond BASELINE cfg 3 (the reference trains on all ~1e5 in-mask pixels of a view per step, bear.conf:47,
in split chunks): ms/step, ray-samples/s and peak device memory per pixel count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
for px in (32768, 65536, 131072, 262144):
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    step = bench.make_step(dev)
    inp, gt = stage2_inputs(px, 96, 8, seed=100, device=dev, with_surface_idx=True)
    l_slt = torch.arange(96, device=dev) + 288
    ns = int(inp['surface_mask'].sum())
    for _ in range(3): step.step(inp, gt, l_slt, train_order=False)
    bench.settle_gc()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 6
    for _ in range(n): step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    print('%7d px (%d surface): %.2f ms/step, %.1f M ray-samples/s, peak memory %.1f GB' % (px, ns, ms, ns * 96 / ms / 1e3, torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del step, inp, gt
