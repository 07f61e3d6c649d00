"""Where do the 400 us around the psn_gemm_tn_grouped launch go?  Wall and per-thread CPU time of the segments of gemm_tn_grouped
(entry -> _Prof ctor -> C call -> _Prof exit -> return) when called from VisibilityPair.backward."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from psnerf_amd import ops, hip
from psnerf_amd.synthetic import stage2_inputs
W, C = time.perf_counter, time.thread_time
marks = []
class P(hip._Prof):
    def __init__(self, *a, **k):
        marks.append(('ctor', W(), C())); super().__init__(*a, **k)
    def __enter__(self):
        r = super().__enter__(); marks.append(('entered', W(), C())); return r
    def __exit__(self, *exc):
        marks.append(('exit0', W(), C())); r = super().__exit__(*exc); marks.append(('exit1', W(), C())); return r
hip._Prof = P
orig = hip.gemm_tn_grouped
seg = collections.defaultdict(lambda: [0.0, 0.0])
on = [False]
def g(*a, **k):
    if not on[0]:
        return orig(*a, **k)
    del marks[:]
    marks.append(('call', W(), C()))
    r = orig(*a, **k)
    marks.append(('ret', W(), C()))
    for (n0, w0, c0), (n1, w1, c1) in zip(marks[:-1], marks[1:]):
        s = seg[n0 + '->' + n1]; s[0] += w1 - w0; s[1] += c1 - c0
    return r
hip.gemm_tn_grouped = g
f = ops.VisibilityPair.backward
def b(*a, **k):
    on[0] = True
    try: return f(*a, **k)
    finally: on[0] = False
ops.VisibilityPair.backward = staticmethod(b)
dev = torch.device('cuda:0')
print('cpus allowed', len(os.sched_getaffinity(0)), 'threads', torch.get_num_threads())
step = bench.make_step(dev)
inp, gt = stage2_inputs(1024, 96, 8, seed=100, device=dev, with_surface_idx=True)
l_slt = torch.arange(96, device=dev) + 288
for _ in range(5): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize(); seg.clear()
for _ in range(100): step.step(inp, gt, l_slt, train_order=False)
torch.cuda.synchronize()
for k, (w, c) in seg.items():
    print('%-18s wall %7.1f us/step   thread cpu %7.1f us/step' % (k, w * 1e4, c * 1e4))
