"""aten ops of one stage-2 train step grouped by input shape (side-stream overlap off, so durations are not stretched)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
step = bench.make_step(dev)
step.model.overlap_small_nets = False
inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
for _ in range(3):
    step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = e.self_cuda_time_total
    if e.key.startswith('aten::') and t >= 1:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:120]))
rows.sort(reverse=True)
print('total %.3f ms' % (sum(r[0] for r in rows) / 1e3))
for t, c, k, sh in rows[:60]:
    print('%8.1f us %3d  %-22s %s' % (t, c, k, sh))
