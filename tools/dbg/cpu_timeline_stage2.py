"""Host-side timeline of one stage-2 step (no extra syncs): when does the host issue each phase?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import bench
from psnerf_amd.synthetic import stage2_inputs
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(32768, 96, 8, seed=100, device=dev)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(3):
    step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
T = []
def mark(name):
    T.append((name, time.perf_counter()))
def wrap(obj, attr, label):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(label + ' >')
        r = f(*a, **k)
        mark(label + ' <')
        return r
    setattr(obj, attr, g)
from psnerf_amd import ops, hip as _hip
wrap(ops.VisibilityPair, 'launch', 'vis launch')
wrap(_hip, 'mlp_infer', 'hip.mlp_infer')
wrap(_hip, 'sg_shade_fwd', 'sg_shade_fwd')
wrap(_hip, 'gemm', 'hip.gemm')
wrap(_hip, 'pe_encode', 'pe_encode')
self = step
mark('start')
mi = dict(inp)
mi['light_direction'] = F.normalize(self.light_para(l_slt), p=2, dim=-1)
mi['light_intensity'] = self.light_inten_para(l_slt)
count = self.dp.global_count(mi['surface_mask'] & mi['object_mask'])
mark('count (sync)')
out = self.model(mi)
mark('forward issued')
terms = dict(self.loss(out, gt, mi, count=count))
terms_n = self.loss_n(out, count=count)
loss = terms['loss'] + terms_n['loss']
mark('losses issued')
self.sg_optimizer.zero_grad(); self.light_optimizer.zero_grad()
loss.backward()
mark('backward issued')
self.sg_optimizer.step()
mark('Adam issued')
self.light_optimizer.step()
mark('SparseAdam issued')
torch.cuda.synchronize()
mark('GPU done')
t0 = T[0][1]
for n, t in T:
    print('%-20s %8.3f ms' % (n, (t - t0) * 1e3))
