"""aten ops of one stage-1 train step grouped by input shape: which torch-side ops touch the big [Q, *] tensors?"""
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
tr = Trainer(Renderer(net, cfg, device=dev), torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
bd = {k: v.to(dev) for k, v in batch.items()}
for _ in range(2):
    tr.train_step(bd, it=6000)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(bd, it=6000)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = e.self_cuda_time_total
    if e.key.startswith('aten::') and t >= 15:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print('total %.3f ms' % (sum(r[0] for r in rows) / 1e3))
for t, c, k, sh in rows[:45]:
    print('%8.1f us %3d  %-22s %s' % (t, c, k, sh))
