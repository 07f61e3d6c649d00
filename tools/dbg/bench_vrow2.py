"""V-row backward chain: chain engine (mlp_infer_kernel<true,16>) vs the two-row-groups prototype (relu_chain2_kernel).
Needs tools/dbg/experiments/relu_chain2_two_row_groups.patch applied to csrc/mlp_infer.hip (PSN_RELU_CHAIN2=0/1 selects the
kernel); without it both columns time the chain engine."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns = 29487
Q = int(os.environ.get('ROWS', 8 * Ns))
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
H = [torch.randn(Q, 256, device=dev) for _ in range(8)]
g = torch.randn(Q, 1, device=dev)
wl = ws[-1].contiguous()
chain = fused.pack_relu_bwd(ws, 3)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
best = {}
for rep in range(3):
    for mode in ('0', '1'):
        os.environ['PSN_RELU_CHAIN2'] = mode
        DZ = [torch.empty(Q, 256, device=dev) for _ in range(8)]
        fn = lambda: chain(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), mask=H, save=DZ)
        best[mode] = min(best.get(mode, 1e9), t(fn))
        res[mode] = [d.clone() for d in DZ]
print('rows %d: chain engine %.3f ms, two-group kernel %.3f ms' % (Q, best['0'], best['1']))
print('bit-identical dumps:', all(torch.equal(a, b) for a, b in zip(res['0'], res['1'])))
