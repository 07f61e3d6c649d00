"""Is the bimodal host time of the eager step the cyclic garbage collector?  Times 100 steps with gc on / off / frozen and counts
collections (gc.callbacks) per generation with their wall time."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
stat = {0: [0, 0.0], 1: [0, 0.0], 2: [0, 0.0]}
t0 = [0.0]
def cb(phase, info):
    if phase == 'start':
        t0[0] = time.perf_counter()
    else:
        s = stat[info['generation']]; s[0] += 1; s[1] += time.perf_counter() - t0[0]
        s.append(info['collected']) if len(s) < 8 else None
gc.callbacks.append(cb)
def run(n=100):
    for k in stat: stat[k] = [0, 0.0]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step.step(inp, gt, l_slt, train_order=False)
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return h / n * 1e3
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
for mode in ('on', 'off', 'on', 'freeze', 'on'):
    if mode == 'off': gc.disable()
    elif mode == 'freeze': gc.enable(); gc.collect(); gc.freeze()
    else: gc.enable(); gc.unfreeze() if hasattr(gc, 'unfreeze') else None
    h = run()
    print('gc %-6s host %.3f ms/step   collections per 100 steps: %s   objects tracked %d' % (
        mode, h, {k: (v[0], round(v[1] * 1e3, 2), v[2:]) for k, v in stat.items()}, len(gc.get_objects())))
