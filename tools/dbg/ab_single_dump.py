"""A/B of the single-dump softplus chains: ops.GeoFieldFused with the value pass dumping (A_l, S_l) per layer
(baseline) vs A_l only, the consumer chains re-forming sigmoid(100 z) = 1 - exp(-100 a) (PSN_ACT_*_A).  Per-chain HIP-event times at
Q points, results compared (outputs and every parameter gradient)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip, ops, fused

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
p = (torch.rand(Q, 3, device=dev) - 0.5).requires_grad_()
names = ['F1 value', 'F2 sweep', 'B1 sweep-adj', 'B2 value-adj']
res, out = {}, {}
for mode in ('two_dumps', 'single_dump', 'two_dumps', 'single_dump'):
    chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39, single_dump=(mode == 'single_dump'))
    best = None
    for it in range(4):
        for q in params:
            q.grad = None
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
        t = [a.elapsed_time(b) for (nm, rows, a, b, _f) in ev][:4] + [e0.elapsed_time(e1), e1.elapsed_time(e2)]
        if it >= 1:
            best = t if best is None else [min(x, y) for x, y in zip(best, t)]
    res.setdefault(mode, []).append(best)
    out[mode] = (logit.detach().clone(), feat.detach().clone(), grad.detach().clone(), [q.grad.detach().clone() for q in params])
rep = {'points': Q}
for mode, runs in res.items():
    b = [min(r[i] for r in runs) for i in range(6)]
    rep[mode] = dict(zip(names + ['forward_ms', 'backward_ms'], [round(x, 3) for x in b]))
    rep[mode]['chains_ms'] = round(sum(b[:4]), 3)
a, b = out['two_dumps'], out['single_dump']
rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
rep['max_rel_diff'] = {'logit': rel(b[0], a[0]), 'feat': rel(b[1], a[1]), 'grad': rel(b[2], a[2]),
                       'param_grads_worst': max(rel(x, y) for x, y in zip(b[3], a[3]))}
print(json.dumps(rep))
