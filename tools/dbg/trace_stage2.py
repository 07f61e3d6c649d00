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
    step.step(inp, gt, l_slt, train_order=False)  # two steps queued without a host synchronisation: the second is reported
    step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name != 'CPU']
evs.sort(key=lambda e: e.time_range.start)
adam = [i for i, e in enumerate(evs) if 'row_adam' in e.name]
evs = evs[adam[0] + 1:]
t0 = evs[0].time_range.start
tot, prev_end, gaps = 0.0, t0, 0.0
for e in evs:
    d = e.time_range.end - e.time_range.start
    gap = e.time_range.start - prev_end
    if gap > 0:
        gaps += gap
    tot += d
    if os.environ.get('UNTIL_VIS') == '1' and e.time_range.start - t0 > 450:
        break
    if d >= float(os.environ.get('MIN_US', '8')) or gap >= 15:
        print('%9.1f  gap %7.1f  %8.1f us  %s' % (e.time_range.start - t0, gap, d, e.name[:100]))
    prev_end = max(prev_end, e.time_range.end)
print('kernels %d  busy %.3f ms  idle %.3f ms  span %.3f ms' % (len(evs), tot / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
