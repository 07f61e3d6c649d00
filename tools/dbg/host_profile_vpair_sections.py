"""Wall time of the calls made inside ops.VisibilityPair.backward (autograd thread), by wrapping the functions it uses."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops, hip, fused
from psnerf_amd.synthetic import stage2_inputs
acc = collections.defaultdict(float); cnt = collections.Counter()
inside = [False]
def wrap(mod, name, tag=None):
    f = getattr(mod, name)
    def g(*a, **k):
        if not inside[0]:
            return f(*a, **k)
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[tag or name] += time.perf_counter() - t0; cnt[tag or name] += 1
    setattr(mod, name, g)
for n in ('gemm_tn_grouped', 'pair_sums_group', 'colsum', 'mlp_infer', 'mlp_pack_layers', 'gemm', '_tn_is_big', '_tn_is_tall', '_tn_aligned', 'workspace', '_mat_ptr', '_ld'):
    wrap(hip, n)
wrap(hip._lib, 'psn_gemm_tn_grouped', 'C psn_gemm_tn_grouped')
wrap(hip._lib, 'psn_mlp_infer', 'C psn_mlp_infer')
wrap(fused, 'pack_relu_bwd'); wrap(torch, 'cat', 'torch.cat'); wrap(torch, 'empty', 'torch.empty')
f = ops.VisibilityPair.backward
def g(*a, **k):
    inside[0] = True
    t0 = time.perf_counter()
    try:
        return f(*a, **k)
    finally:
        acc['TOTAL backward'] += time.perf_counter() - t0; inside[0] = False
ops.VisibilityPair.backward = staticmethod(g)
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); acc.clear(); cnt.clear()
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print('%-22s %8.1f us/step  (%d calls/step)' % (k, v * 1e4, cnt[k] // 100))
