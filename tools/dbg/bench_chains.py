"""Per-chain timing of ops.GeoFieldFused (value pass, sweep, and the two adjoint chains) at one chunk size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip, ops, fused

Q = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 262144
X3 = 'x3' in sys.argv          # split-bf16 weight stages (PSN_W_BF16X2), on top of the single-dump chains
SINGLE = 'single' in sys.argv or X3  # the single-dump experiment (ops.GEO_SINGLE_DUMP): consumer chains re-form the sigmoid from A
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
p = (torch.rand(Q, 3, device=dev) - 0.5).requires_grad_()
chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39, single_dump=SINGLE, x3=X3)
names = ['F1 value', 'F2 sweep', 'B1 sweep-adj', 'B2 value-adj']
macs = [39 * 256 + 6 * 65536 + 256 * 217 + 256 * 256 + 2 * 65536 + 256, 7 * 65536 + 256 * 64, 64 * 256 + 7 * 65536 + 64 * 256, 8 * 65536]
for it in range(3):
    hip.PROFILE_EVENTS = []
    torch.cuda.synchronize()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e0.record()
    logit, feat, grad = ops.GeoFieldFused.apply(p, 6, 1.0, (4,), True, chains, None, *params)
    e1.record()
    (logit.sum() + feat.sum() * 0.1 + (grad * grad).sum()).backward()
    e2.record()
    torch.cuda.synchronize()
    ev = hip.PROFILE_EVENTS
    hip.PROFILE_EVENTS = None
    if it == 2:
        for n, m, (nm, rows, a, b, _f) in zip(names, macs, ev):
            ms = a.elapsed_time(b)
            print('%-14s %7.3f ms  %6.1f TF/s (padded-256 MACs)' % (n, ms, 2.0 * m * rows / ms / 1e9))
        print('forward %.3f ms  backward %.3f ms' % (e0.elapsed_time(e1), e1.elapsed_time(e2)))

if os.environ.get('CHAINS_ONLY') or SINGLE or 'chains-only' in sys.argv:
    sys.exit(0)
# ---- where does the value chain's time go?  same launch with fewer dumps
pe = hip.pe_encode(p.detach(), 6, 64, 1.0)
A = [torch.empty(Q, 256, device=dev) for _ in range(8)]
S = [torch.empty(Q, 256, device=dev) for _ in range(8)]
feat = torch.empty(Q, 256, device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print('F1  9 dumps   %.3f ms' % timeit(lambda: chains['fwd'](pe, Q, save=A + [feat])))
print('F1  1 dump    %.3f ms' % timeit(lambda: chains['fwd'](pe, Q, save=[None] * 8 + [feat])))
U = [torch.empty(Q, 256, device=dev) for _ in range(8)]
R = [torch.empty(Q, 256, device=dev) for _ in range(8)]
w_row = params[16][0:1].detach().contiguous()
print('F2 8 loads 16 dumps %.3f ms' % timeit(lambda: chains['sweep'](None, Q, a_div=1, a_mod=1, init_a_direct=w_row, mask=S + [None], save=U + [feat], save2=[None] + R[1:] + [None])))
print('F2 8 loads  1 dump  %.3f ms' % timeit(lambda: chains['sweep'](None, Q, a_div=1, a_mod=1, init_a_direct=w_row, mask=S + [None], save=[None] * 8 + [feat])))
