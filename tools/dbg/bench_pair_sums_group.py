"""psn_pair_sums_group at the BEAR step's shape (2 input layers, V = 8, Ns = 29487 / 3686, C = 256): us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip
dev = torch.device('cuda:0')
for Ns in (29487, 3686):
    V, C = 8, 256
    xs = [torch.randn(V * Ns, C, device=dev) for _ in range(2)]
    pl = torch.randn(V, 64, device=dev)
    for _ in range(3): hip.pair_sums_group(xs, V, Ns, pl, 64, [True, False])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r = hip.pair_sums_group(xs, V, Ns, pl, 64, [True, False])
    e1.record(); torch.cuda.synchronize()
    ref = xs[0].view(V, Ns, C).double().sum(1).t() @ pl.double()
    err = float((r[0][1][:, :64].double() - ref).abs().max() / ref.abs().max())
    print('Ns %6d: %.1f us per call, dW_l rel err %.1e' % (Ns, e0.elapsed_time(e1) / 20 * 1e3, err))
