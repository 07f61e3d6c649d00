"""Input-block weight gradients (256 x 39 / 33 outputs) of a stage-1 step: kernel + reduction time (the K-slice count was
swept 192 ... 1024 through a temporary environment override: 256 - 512 slices in total are best)."""
import os, sys, subprocess
if len(sys.argv) == 1:
    sys.argv.append('default')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
dev = torch.device('cuda')
Q = 524288
g = torch.Generator(device=dev).manual_seed(0)
dZ, U, dz0 = [torch.randn(Q, 256, device=dev, generator=g) for _ in range(3)]
pe, ddpe, x = [torch.randn(Q, 64, device=dev, generator=g) for _ in range(3)]
big = [dict(A=dZ, B=U, colsum=True)]
sets = {'geo l=0 (2 x N=39)': [dict(A=dZ, B=pe[:, :39], A2=U, B2=ddpe[:, :39], colsum=True)],
        'app (N=33)': [dict(A=dz0, B=x[:, :33], colsum=True)],
        'one 256x256 for scale': big}
for name, items in sets.items():
    for _ in range(3): hip.gemm_tn_grouped(items)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.gemm_tn_grouped(items)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = sum((it['A'].numel() + it['B'].shape[0] * it['B'].shape[1]) * 4 * (2 if 'A2' in it else 1) for it in items)
    print('slices %-5s %-24s %.3f ms  (%.2f TB/s of operand bytes)' % (sys.argv[1], name, ms, by / ms * 1e-9))
