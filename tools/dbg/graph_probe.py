"""Probe: does a whole stage-2 TrainStep.step capture into a HIP graph as it is (host scalars baked in)?  Timing only.
usage: graph_probe.py PIXELS OVERLAP(0/1)"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
px = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
step = bench.make_step(dev)
step.model.overlap_small_nets = bool(overlap)
inp, gt = stage2_inputs(px, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4):
        step.step(inp, gt, l_slt, train_order=False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
def timeit(fn, k=40):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(6e8)); e0.record()
    for _ in range(k): fn()
    e1.record(); th = time.perf_counter(); torch.cuda.synchronize()
    return (t2 - t0) / k * 1e3, (t1 - t0) / k * 1e3, e0.elapsed_time(e1) / k, (time.perf_counter() - th) * 1e3
print('eager px=%d overlap=%d wall %.3f ms host %.3f ms gpu %.3f ms (drain wait %.1f ms)' % ((px, overlap) + timeit(lambda: step.step(inp, gt, l_slt, train_order=False))), flush=True)
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        terms, out = step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    print('graph px=%d overlap=%d wall %.3f ms host %.3f ms gpu %.3f ms (drain wait %.1f ms)' % ((px, overlap) + timeit(g.replay)), flush=True)
except Exception:
    traceback.print_exc()
