"""Cache-resident chunking of the stage-1 geometry chains: the same Q query points through
ops.GeoFieldFused forward (F1 value pass, F2 sweep) + backward (B1, B2, grouped weight gradients, accumulating) in chunks of
C rows, every chunk's backward right behind its forward, so that a chunk's dumps (48 KB per point) are re-read while they
may still be in the 256 MB Infinity Cache.  Prints ms per Q points for each chunk size; with PMC=1 runs one chunk size only
(CHUNK env) for the counter passes of tools/dbg/pmc_chains_chunked.sh."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip, ops, fused

Q = int(os.environ.get('Q', 262144))
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
p_all = (torch.rand(Q, 3, device=dev) - 0.5)
chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39)


def run(C):
    for q in params:
        q.grad = None
    for s in range(0, Q, C):
        p = p_all[s:s + C].clone().requires_grad_()
        logit, feat, grad = ops.GeoFieldFused.apply(p, 6, 1.0, (4,), True, chains, None, *params)
        (logit.sum() + feat.sum() * 0.1 + (grad * grad).sum()).backward()


def timeit(C, n=3):
    run(C)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        run(C)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


if os.environ.get('PMC') == '1':
    C = int(os.environ.get('CHUNK', Q))
    run(C)
    run(C)
    torch.cuda.synchronize()
    sys.exit(0)

res = {}
ref_grads = None
for C in (Q, 131072, 65536, 32768, 16384, 8192, 4096):
    ms = timeit(C)
    # chain-only time of the last run from HIP events
    hip.PROFILE_EVENTS = ev = []
    run(C)
    torch.cuda.synchronize()
    hip.PROFILE_EVENTS = None
    by = {}
    for (nm, rows, a, b, _f) in ev:
        by[nm] = by.get(nm, 0.0) + a.elapsed_time(b)
    g = torch.cat([q.grad.reshape(-1) for q in params])
    if ref_grads is None:
        ref_grads = g.clone()
    err = float((g - ref_grads).abs().max() / ref_grads.abs().max())
    res[C] = {'ms_total': round(ms, 3), 'launch_ms_by_kind': {k: round(v, 3) for k, v in by.items()}, 'n_chunks': Q // C,
              'grad_rel_diff_vs_unchunked': err}
    print('chunk %7d rows x %3d: %8.3f ms per %d points   %s   grad diff %.1e' % (C, Q // C, ms, Q, res[C]['launch_ms_by_kind'], err), flush=True)
print(json.dumps(res))
