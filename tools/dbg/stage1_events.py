"""Per-launch HIP-event durations of one stage-1 train step (bench configuration)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
from psnerf_amd.synthetic import stage1_batch, stage1_cfg
dev = torch.device('cuda:0')
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
batch = {k: v.to(dev) for k, v in stage1_batch(cfg, h=512, w=612, seed=0).items()}
torch.manual_seed(42)
net = NeuralNetwork(cfg)
tr = Trainer(Renderer(net, cfg, device=dev), torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
for _ in range(3):
    tr.train_step(batch, it=6000)
torch.cuda.synchronize()
hip.PROFILE_EVENTS = ev = []
tr.train_step(batch, it=6000)
torch.cuda.synchronize()
hip.PROFILE_EVENTS = None
tot = 0.0
for (k, u, a, b, f) in ev:
    ms = a.elapsed_time(b); tot += ms
    print('%-18s units %-10s %8.3f ms %s' % (k, u if not isinstance(u, float) else '%.3g' % u, ms, ('%.1f TF' % (f / ms * 1e-9)) if f else ''))
print('sum of instrumented launches %.3f ms' % tot)
