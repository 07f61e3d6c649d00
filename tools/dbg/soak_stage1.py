"""Stage-1 determinism soak: two identical runs of N train steps (same seeds) must end with bit-identical parameters --
catches races of the asynchronous pieces (pinned pixel upload, prefetched march, packs built ahead of their use)."""
import os, sys, hashlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd.synthetic import stage1_cfg, stage1_batch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
batch = stage1_batch(cfg, h=512, w=612, seed=0)
dev = torch.device('cuda:0')
out = []
for run in range(2):
    torch.manual_seed(42)
    net = NeuralNetwork(cfg)
    from psnerf_amd.optim import FlatAdam
    ren = Renderer(net, cfg, device=dev)
    tr = Trainer(ren, FlatAdam(net.parameters(), lr=1e-4), cfg, device=dev)
    bd = {k: v.to(dev) for k, v in batch.items()}
    torch.manual_seed(7)
    for i in range(steps):
        terms = tr.train_step(bd, it=6000 + i)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, v in sorted(net.state_dict().items()):
        h.update(v.detach().cpu().numpy().tobytes())
    out.append((float(terms['loss'].detach()), h.hexdigest()[:16]))
    print('run', run, out[-1], flush=True)
print('bit-identical:', out[0] == out[1])
