"""torch.profiler view of one stage-1 train step grouped by python call site: where do the small torch kernels come from?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from psnerf_amd.synthetic import stage1_cfg, stage1_batch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
batch = stage1_batch(cfg, h=512, w=612, seed=0)
dev = torch.device('cuda:0')
torch.manual_seed(42)
net = NeuralNetwork(cfg)
ren = Renderer(net, cfg, device=dev)
tr = Trainer(ren, torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
bd = {k: v.to(dev) for k, v in batch.items()}
for _ in range(2):
    tr.train_step(bd, it=6000)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(bd, it=6000)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = e.self_cuda_time_total
    if not e.key.startswith('aten::') or t < 8:
        continue
    st = [s for s in e.stack if 'psnerf_amd' in s or 'torch/optim' in s][:3]
    rows.append((t, e.count, e.key, ' <- '.join(s.split('/')[-1] for s in st)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('aten ops with a psnerf_amd frame: %.3f ms' % (tot / 1e3))
for t, c, k, st in rows[:60]:
    print('%8.1f us %4d  %-22s %s' % (t, c, k, st[:170]))
