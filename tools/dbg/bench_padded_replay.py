"""Replayed 4096-px step: exact surface list (adopt_inputs) vs list padded to the pixel count, padding evaluated / skipped."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
from psnerf_amd.stage2.graph import GraphedTrainStep
dev = torch.device('cuda:0')
px = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
def run(mode):
    step = bench.make_step(dev)
    inp, gt = stage2_inputs(px, 96, 8, seed=100, device=dev, with_surface_idx=(mode == 'exact'))
    l_slt = torch.arange(96, device=dev) + 288
    GraphedTrainStep.SKIP_PADDING = mode != 'pad_eval'
    if mode.startswith('mult'):
        inp['surface_idx'] = inp['surface_mask'][0].nonzero(as_tuple=True)[0]
        r = GraphedTrainStep(step, pad_multiple=int(mode[4:]))
    else:
        r = GraphedTrainStep(step, adopt_inputs=True) if mode == 'exact' else GraphedTrainStep(step, pad_to_pixels=True)
    for _ in range(6): r.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(50): r.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 50 * 1e3)
    assert r.n_replays >= 150
    return best
for mode in ('exact', 'pad_eval', 'pad_skip', 'mult64', 'mult256', 'mult512', 'exact', 'pad_skip', 'mult64', 'mult256'):
    print('%-9s %.3f ms/step' % (mode, run(mode)), flush=True)
