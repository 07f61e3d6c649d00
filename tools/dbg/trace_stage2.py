"""Kernel timeline (launch order, duration) of one stage-2 train step, side-stream overlap off."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
step = bench.make_step(dev)
step.model.overlap_small_nets = os.environ.get('OVERLAP', '0') == '1'
inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
for _ in range(3):
    step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name != 'CPU']
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
tot = 0.0
for e in evs:
    d = e.time_range.end - e.time_range.start
    tot += d
    if d >= float(os.environ.get('MIN_US', '8')):
        print('%9.1f  %8.1f us  %s' % (e.time_range.start - t0, d, e.name[:110]))
print('kernels %d  busy %.3f ms  span %.3f ms' % (len(evs), tot / 1e3, (evs[-1].time_range.end - t0) / 1e3))
