"""Eager (default) or replayed stage-2 step for a rocprofv3 --kernel-trace timeline: python trace_step.py PIXELS [graph]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
from psnerf_amd.stage2.graph import GraphedTrainStep
dev = torch.device('cuda:0')
px = int(sys.argv[1])
step = bench.make_step(dev)
inp, gt = stage2_inputs(px, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
r = GraphedTrainStep(step, adopt_inputs=True) if len(sys.argv) > 2 else step
for _ in range(12): r.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
