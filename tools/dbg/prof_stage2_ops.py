"""torch.profiler view of one stage-2 train step (bench configuration): which aten ops emit the small kernels."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev)
l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
for _ in range(3):
    step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=len(sys.argv) > 1) as prof:
    step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
if len(sys.argv) > 1:  # group the small aten ops by python call site
    print(prof.key_averages(group_by_stack_n=4).table(sort_by='cuda_time_total', row_limit=70, max_name_column_width=40, max_src_column_width=110))
else:
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=60, max_name_column_width=60))
