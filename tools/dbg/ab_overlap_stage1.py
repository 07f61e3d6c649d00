"""Stage-1 train step with / without the normals on a side stream, same process, alternating blocks."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
from psnerf_amd.synthetic import stage1_batch, stage1_cfg
dev = torch.device('cuda:0')
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
batch = {k: v.to(dev) for k, v in stage1_batch(cfg, h=512, w=612, seed=0).items()}
torch.manual_seed(42)
net = NeuralNetwork(cfg)
ren = Renderer(net, cfg, device=dev)
tr = Trainer(ren, torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
for _ in range(3):
    tr.train_step(batch, it=6000)
best = {True: 1e9, False: 1e9}
for rep in range(4):
    for ov in (True, False):
        ren.overlap_normals = ov
        tr.train_step(batch, it=6000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            tr.train_step(batch, it=6000)
        torch.cuda.synchronize()
        best[ov] = min(best[ov], (time.perf_counter() - t0) / 8 * 1e3)
print('stage-1 step: normals on a side stream %.3f ms, on the main stream %.3f ms' % (best[True], best[False]))
