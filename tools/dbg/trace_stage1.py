"""Kernel timeline (launch order, duration, idle gap before) of one stage-1 train step; two steps are queued without a host
synchronisation in between and the SECOND is reported, so the host has run ahead as in a training loop."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from torch.profiler import profile, ProfilerActivity
from psnerf_amd.synthetic import stage1_cfg, stage1_batch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': int(os.environ.get('RAYS', '4096'))})
batch = stage1_batch(cfg, h=512, w=612, seed=0)
dev = torch.device('cuda:0')
torch.manual_seed(42)
net = NeuralNetwork(cfg)
tr = Trainer(Renderer(net, cfg, device=dev), torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
bd = {k: v.to(dev) for k, v in batch.items()}
for _ in range(3):
    tr.train_step(bd, it=6000)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.train_step(bd, it=6000)
    tr.train_step(bd, it=6000)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name != 'CPU']
evs.sort(key=lambda e: e.time_range.start)
# second step = from the second 'sample_points' launch on (the first kernel family of a step that occurs once)
names = [e.name for e in evs]
first = [i for i, n in enumerate(names) if 'first_crossing' in n]
cut = 0
if len(first) >= 2:
    # walk back from the second march to the start of that step: the step starts with the kernel after the previous Adam
    adam = [i for i, n in enumerate(names) if 'multi_tensor_apply' in n and i < first[1]]
    cut = adam[-1] + 1 if adam else 0
evs = evs[cut:]
t0 = evs[0].time_range.start
busy, prev_end, gaps = 0.0, t0, 0.0
small = 0.0
MIN = float(os.environ.get('MIN_US', '15'))
for e in evs:
    d = e.time_range.end - e.time_range.start
    gap = max(0.0, e.time_range.start - prev_end)
    gaps += gap
    busy += d
    if 'psn::' not in e.name:
        small += d
    if d >= MIN or gap >= 20:
        print('%9.1f  gap %6.1f  %8.1f us  %s' % (e.time_range.start - t0, gap, d, e.name[:100]))
    prev_end = max(prev_end, e.time_range.end)
print('kernels %d  busy %.3f ms  idle gaps %.3f ms  span %.3f ms  non-psn kernels %.3f ms' % (len(evs), busy / 1e3, gaps / 1e3, (prev_end - t0) / 1e3, small / 1e3))
if os.environ.get('NONPSN') == '1':
    agg = {}
    for e in evs:
        if 'psn::' not in e.name:
            d = e.time_range.end - e.time_range.start
            k = e.name[:150]
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%4d  %8.1f us  %s' % (c, d, k))
if os.environ.get('NONPSN_LIST') == '1':
    prev = ''
    for e in evs:
        d = e.time_range.end - e.time_range.start
        if 'psn::' in e.name:
            prev = e.name[:40]
        elif d >= 9:
            print('%8.1f us  after %-40s %s' % (d, prev, e.name[:120]))
