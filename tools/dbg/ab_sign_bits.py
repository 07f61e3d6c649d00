"""A/B: ReLU-backward chains on sign-bit words (ops.RELU_SIGN_BITS) vs the activation rows as masks; eager 32768 px, replayed 4096 px."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops
from psnerf_amd.synthetic import stage2_inputs
from psnerf_amd.stage2.graph import GraphedTrainStep
dev = torch.device('cuda:0')
def run(px, on, graph):
    ops.RELU_SIGN_BITS = on
    step = bench.make_step(dev)
    inp, gt = stage2_inputs(px, 96, 8, seed=100, device=dev, with_surface_idx=True)
    l_slt = torch.arange(96, device=dev) + 288
    r = GraphedTrainStep(step, adopt_inputs=True) if graph else step
    n = 60 if px <= 8192 else 12
    for _ in range(6): terms, _ = r.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(n): terms, _ = r.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best, float(terms['total'].detach())
for px, graph in ((32768, False), (4096, True), (32768, False), (4096, True)):
    for on in (False, True):
        print('px %5d %s sign_bits=%s  %.3f ms/step  loss %.9f' % ((px, 'graph' if graph else 'eager', on) + run(px, on, graph)), flush=True)
