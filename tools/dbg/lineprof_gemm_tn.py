"""Per-line wall time inside hip.gemm_tn_grouped when called from ops.VisibilityPair.backward (autograd thread), via sys.settrace."""
import os, sys, time, collections, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops, hip
from psnerf_amd.synthetic import stage2_inputs
code = hip.gemm_tn_grouped.__code__
acc = collections.defaultdict(float); hits = collections.Counter()
last = [None, 0.0]
def tracer(frame, event, arg):
    if frame.f_code is not code:
        return None
    def local(frame, event, arg):
        now = time.perf_counter()
        if last[0] is not None:
            acc[last[0]] += now - last[1]; hits[last[0]] += 1
        last[0], last[1] = (frame.f_lineno if event != 'return' else None), now
        return local
    last[0], last[1] = frame.f_lineno, time.perf_counter()
    return local
orig = hip.gemm_tn_grouped
on = [False]
tot = [0.0]
def g(*a, **k):
    if not on[0]:
        return orig(*a, **k)
    sys.settrace(tracer)
    t0 = time.perf_counter()
    try:
        return orig(*a, **k)
    finally:
        tot[0] += time.perf_counter() - t0
        sys.settrace(None); last[0] = None
hip.gemm_tn_grouped = g
f = ops.VisibilityPair.backward
def b(*a, **k):
    on[0] = True
    try: return f(*a, **k)
    finally: on[0] = False
ops.VisibilityPair.backward = staticmethod(b)
dev = torch.device('cuda:0')
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); acc.clear(); hits.clear(); tot[0] = 0
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
src, first = inspect.getsourcelines(orig)
print('gemm_tn_grouped total %.1f us/step' % (tot[0] * 1e4))
for ln, v in sorted(acc.items(), key=lambda x: -x[1])[:12]:
    print('%7.1f us/step  %5d hits/step  L%d: %s' % (v * 1e4, hits[ln] // 100, ln, src[ln - first].strip()[:110]))
