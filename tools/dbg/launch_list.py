"""Every device kernel launch of ONE steady-state train step in launch order, with the aten op and the psnerf_amd call site
that issued it (stage 2 by default, `stage1` as argument).  For hunting the torch-eager tail."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
if len(sys.argv) > 1 and sys.argv[1] == 'stage1':
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.optim import FlatAdam
    from psnerf_amd.synthetic import stage1_batch, stage1_cfg
    cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
    batch = {k: v.to(dev) for k, v in stage1_batch(cfg, h=512, w=612, seed=0).items()}
    torch.manual_seed(42)
    net = NeuralNetwork(cfg)
    ren = Renderer(net, cfg, device=dev)
    tr = Trainer(ren, FlatAdam(net.parameters(), lr=1e-4), cfg, device=dev)
    run = lambda: tr.train_step(batch, it=6000)
else:
    from psnerf_amd.synthetic import stage2_inputs
    step = bench.make_step(dev)
    inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev, with_surface_idx=True)
    l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
    run = lambda: step.step(inp, gt, l_slt, train_order=False)
for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    # innermost op only: skip an op one of whose children carries the same kernels
    if any(getattr(c, 'kernels', None) for c in (e.cpu_children or [])):
        continue
    site = ''
    for fr in (e.stack or []):
        if 'psnerf_amd' in fr or 'bench.py' in fr:
            site = fr.split('psnerf_amd/')[-1] if 'psnerf_amd/' in fr else fr
            break
    for k in ks:
        rows.append((e.time_range.start, e.name, k.name[:70], k.duration, site[:90]))
rows.sort()
n_psn = sum(1 for r in rows if 'psn::' in r[2])
print('%d launches, %d hand-written (psn::), %d other' % (len(rows), n_psn, len(rows) - n_psn))
for i, (t, op, kn, dur, site) in enumerate(rows):
    print('%3d %-34s %-70s %7.1f us  %s' % (i, op[:34], kn, dur, site))
