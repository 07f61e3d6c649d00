"""Geometry chains (ops.GeoFieldFused: value pass, sweep, the two adjoints) and appearance chains: exact fp32 vs the split-bf16
form (ops.chain_precision('bf16x3')).  Per-chain kernel time at Q points, and the distance of outputs / parameter gradients of
both forms from a float64 evaluation of the same network (torch autograd, double backward) at a small Q.
    python tools/dbg/ab_chain_x3.py [Q]            -> one JSON line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip, ops, fused

Q = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 262144
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
names = ['F1 value', 'F2 sweep', 'B1 sweep-adj', 'B2 value-adj']
rep = {'points': Q, 'chains_ms': {}, 'accuracy': {}}


def run(pts, x3):
    for q in params:
        q.grad = None
    pts = pts.clone().requires_grad_()
    with ops.chain_precision('bf16x3' if x3 else 'fp32'):
        chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39, single_dump=ops.GEO_SINGLE_DUMP, x3=x3)
        logit, feat, grad = ops.GeoFieldFused.apply(pts, 6, 1.0, (4,), True, chains, None, *params)
        (logit.sum() + feat.sum() * 0.1 + (grad * grad).sum()).backward()
    return logit.detach(), feat.detach(), grad.detach(), [q.grad.clone() for q in params]


p = torch.rand(Q, 3, device=dev) - 0.5
for x3 in (False, True):
    for it in range(3):
        hip.PROFILE_EVENTS = []
        run(p, x3)
        torch.cuda.synchronize()
        ev = hip.PROFILE_EVENTS
        hip.PROFILE_EVENTS = None
    rep['chains_ms']['bf16x3' if x3 else 'fp32'] = {n: round(a.elapsed_time(b), 3) for n, (nm, rows, a, b, _f) in zip(names, ev)}
    rep['chains_ms']['bf16x3' if x3 else 'fp32']['sum'] = round(sum(a.elapsed_time(b) for (nm, rows, a, b, _f) in ev[:4]), 3)

# accuracy against float64 at a small point set
Qs = 4096
ps = (torch.rand(Qs, 3, device=dev) - 0.5)
P64 = [q.detach().double().requires_grad_() for q in params]


def ref64(pts):
    pts = pts.double().requires_grad_()
    enc = [pts] + [f(pts * 2.0 ** k) for k in range(6) for f in (torch.sin, torch.cos)]
    pe = torch.cat(enc, -1)
    h = pe
    for l in range(9):
        W, b = P64[2 * l], P64[2 * l + 1]
        if l == 4:
            h = torch.cat([h, pe], -1)  # (effective weights: the 1 / sqrt(2) of network.py:90-91 is folded in)
        h = h @ W.t() + b
        if l < 8:
            h = torch.nn.functional.softplus(h, beta=100)
    logit, feat = h[:, :1], h[:, 1:]
    g = torch.autograd.grad(logit.sum(), pts, create_graph=True)[0]
    (logit.sum() + feat.sum() * 0.1 + (g * g).sum()).backward()
    return logit.detach(), feat.detach(), g.detach(), [q.grad for q in P64]


r64 = ref64(ps)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


for x3 in (False, True):
    o = run(ps, x3)
    rep['accuracy']['bf16x3' if x3 else 'fp32'] = {
        'logit': rel(o[0], r64[0]), 'feat': rel(o[1], r64[1]), 'grad': rel(o[2], r64[2]),
        'd_params_worst': max(rel(a, b) for a, b in zip(o[3], r64[3])), 'd_params_weights': [round(rel(a, b), 9) for a, b in zip(o[3][0::2], r64[3][0::2])]}
print(json.dumps(rep))
