"""A/B of the chain engine's two forms (psn_mlp_chain_pipeline): the four stage-1 geometry chains (F1 value pass, F2 sweep, B1 sweep
adjoint, B2 value adjoint) at one chunk size, classic (activation program between two layers) vs pipelined (inside the next layer's
stage loop): per-chain HIP-event times, results compared bit for bit (outputs and every parameter gradient).
    python tools/ab_chain_pipe.py [rows] [--mode classic|pipe]     (--mode: one form only, for rocprofv3 --pmc passes)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psnerf_amd import hip, ops, fused

Q = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 262144
only = sys.argv[sys.argv.index('--mode') + 1] if '--mode' in sys.argv else None
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
p = (torch.rand(Q, 3, device=dev) - 0.5).requires_grad_()
chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39, single_dump=ops.GEO_SINGLE_DUMP)
names = ['F1 value', 'F2 sweep', 'B1 sweep-adj', 'B2 value-adj']
macs = [39 * 256 + 6 * 65536 + 256 * 217 + 256 * 256 + 2 * 65536 + 256, 7 * 65536 + 256 * 64, 64 * 256 + 7 * 65536 + 64 * 256, 8 * 65536]
rep, res = {'rows': Q}, {}
for mode in ([only] if only else ['classic', 'pipe', 'classic', 'pipe']):
    with hip.chain_pipeline(mode == 'pipe'):
        for it in range(3):
            for q in params + [p]:
                q.grad = None
            hip.PROFILE_EVENTS = []
            torch.cuda.synchronize()
            logit, feat, grad = ops.GeoFieldFused.apply(p, 6, 1.0, (4,), True, chains, None, *params)
            (logit.sum() + feat.sum() * 0.1 + (grad * grad).sum()).backward()
            torch.cuda.synchronize()
            ev = hip.PROFILE_EVENTS
            hip.PROFILE_EVENTS = None
    ms = [a.elapsed_time(b) for (nm, rows, a, b, _f) in ev[:4]]
    rep.setdefault(mode, []).append({n: round(t, 3) for n, t in zip(names, ms)} | {'sum_ms': round(sum(ms), 3),
                                     'tflops_padded': {n: round(2.0 * m * Q / t / 1e9, 1) for n, m, t in zip(names, macs, ms)}})
    res[mode] = [logit.detach().clone(), feat.detach().clone(), grad.detach().clone()] + [q.grad.clone() for q in params + [p] if q.grad is not None]
if len(res) == 2:
    rep['bit_identical'] = all(torch.equal(a, b) for a, b in zip(res['classic'], res['pipe']))
    rep['max_abs_diff'] = max(float((a - b).abs().max()) for a, b in zip(res['classic'], res['pipe']))
print(json.dumps(rep))
