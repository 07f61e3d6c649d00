"""Host-side timeline of one stage-1 step: where does the host stall (hidden synchronisations)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd.synthetic import stage1_cfg, stage1_batch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
from psnerf_amd import hip as _hip
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': 4096})
batch = stage1_batch(cfg, h=512, w=612, seed=0)
dev = torch.device('cuda:0')
torch.manual_seed(42)
net = NeuralNetwork(cfg)
ren = Renderer(net, cfg, device=dev)
tr = Trainer(ren, torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev)
bd = {k: v.to(dev) for k, v in batch.items()}
for _ in range(3):
    tr.train_step(bd, it=6000)
torch.cuda.synchronize()
T = []
def wrap(obj, attr, label):
    f = getattr(obj, attr)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T.append((label, t, time.perf_counter()))
        return r
    setattr(obj, attr, g)
wrap(_hip, 'mlp_infer', 'mlp_infer')
wrap(_hip, 'gemm_tn_grouped', 'gemm_tn_grouped')
wrap(_hip, 'composite_fwd', 'composite_fwd')
wrap(_hip, 'composite_bwd', 'composite_bwd')
wrap(_hip, 'root_find', 'root_find')
wrap(_hip, 'first_crossing', 'first_crossing')
wrap(_hip, 'sample_points', 'sample_points')
wrap(torch.Tensor, 'nonzero', 'SYNC nonzero')
wrap(torch.Tensor, '__int__', 'SYNC int()')
wrap(torch.Tensor, 'item', 'SYNC item')
wrap(tr.optimizer, 'step', 'adam.step')
t0 = time.perf_counter()
tr.train_step(bd, it=6000)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
prev = t0
for n, a, b in T:
    print('%-18s issued at %7.2f ms (host gap before it %6.2f ms)' % (n, (a - t0) * 1e3, (a - prev) * 1e3))
    prev = b
print('train_step returned at %.2f ms, GPU done at %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
