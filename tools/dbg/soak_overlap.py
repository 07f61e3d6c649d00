"""Soak test of the side-stream overlap at bench size: N steps with and without it must end bit-identical."""
import os, sys, hashlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
out = {}
for overlap in (True, False, True):
    torch.manual_seed(0)
    step = bench.make_step(dev)
    step.model.overlap_small_nets = overlap
    for i in range(steps):
        terms, _ = step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, v in sorted(step.model.state_dict().items()):
        h.update(v.detach().cpu().numpy().tobytes())
    h.update(step.light_para.weight.detach().cpu().numpy().tobytes())
    out.setdefault(overlap, []).append((float(terms['total']), h.hexdigest()[:16]))
    print('overlap', overlap, out[overlap][-1], flush=True)
vals = [v for vs in out.values() for v in vs]
print('bit-identical:', all(v == vals[0] for v in vals))
