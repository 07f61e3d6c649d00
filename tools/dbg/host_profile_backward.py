"""Host time of every autograd node's Python backward / forward in a stage-2 step (tiny batch), accumulated by wrapping ops.*"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops
from psnerf_amd.synthetic import stage2_inputs
acc = collections.defaultdict(float)
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[cls.__name__ + '.' + name] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(g))
for n in dir(ops):
    c = getattr(ops, n)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, 'forward'); wrap(c, 'backward')
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
if os.environ.get('FREEZE') == '1':
    import gc
    gc.collect(); gc.freeze()
elif os.environ.get('FREEZE') == 'off':
    import gc
    gc.disable()
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('host issue per step %.3f ms' % ((t1 - t0) * 10))
for k, v in sorted(acc.items(), key=lambda x: -x[1])[:24]:
    print('%-40s %.3f ms/step' % (k, v * 10))
print('sum of wrapped: fwd %.3f bwd %.3f ms/step' % (sum(v for k, v in acc.items() if k.endswith('forward')) * 10, sum(v for k, v in acc.items() if k.endswith('backward')) * 10))
