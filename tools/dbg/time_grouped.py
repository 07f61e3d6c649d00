"""Per-call timing of hip.gemm_tn_grouped inside one stage-2 train step (bench configuration), with the item shapes."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from psnerf_amd import hip
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev)
l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
for _ in range(3):
    step.step(inp, gt, l_slt, train_order=False)
orig = hip.gemm_tn_grouped
log = []
def wrapped(items, split_k=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(items, split_k)
    e1.record()
    log.append((e0, e1, [(tuple(it['A'].shape), tuple(it['B'].shape), 'A2' in it and it['A2'] is not None, it.get('b_div'), 'B_tab2' in it and it['B_tab2'] is not None) for it in items]))
    return r
hip.gemm_tn_grouped = wrapped
import psnerf_amd.ops as ops
step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
for e0, e1, shapes in log:
    print('%.3f ms' % e0.elapsed_time(e1))
    for s in shapes:
        print('     A %s  B %s  two-product %s  b_div %s  two-table %s' % s)
