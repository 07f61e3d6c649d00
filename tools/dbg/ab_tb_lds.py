"""A/B (separate processes: the switch is read once): the B-side init row of the visibility launch through LDS (PSN_TB_LDS, default)
vs 16 global loads per lane and init layer; eager 32768-px step time, dominant-kernel time, loss digest."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) < 2:
    for mode in ('0', '1', '0', '1'):
        print(subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, PSN_TB_LDS=mode), capture_output=True, text=True).stdout.strip())
    sys.exit(0)
import time, torch
import bench
from psnerf_amd import hip
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(32768, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(4): terms, _ = step.step(inp, gt, l_slt, train_order=False)
bench.settle_gc(); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(12): terms, _ = step.step(inp, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 12 * 1e3)
hip.PROFILE_EVENTS = ev = []
for _ in range(6): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); hip.PROFILE_EVENTS = None
d = [a.elapsed_time(b) for (n, r, a, b, f) in ev if n == 'mlp_infer' and r > 3000000]
print('PSN_TB_LDS=%s: %.3f ms/step, visibility launch %.3f ms (min %.3f), loss %.9f' % (sys.argv[1], best, sum(d) / len(d), min(d), float(terms['total'].detach())))
