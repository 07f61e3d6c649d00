"""psn_app_input on the 524,288 rows of a stage-1 step."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
dev = torch.device('cuda')
Q = 524288
p, v, n = (torch.randn(Q, 3, device=dev) for _ in range(3))
for _ in range(3): hip.app_input(p, v, n, 4)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): hip.app_input(p, v, n, 4)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print('app_input %d rows: %.1f us (%.2f TB/s written)' % (Q, ms * 1e3, Q * 256 / ms / 1e9))
