"""A/B of whole-library builds in ONE process (same box, same clocks): python tools/dbg/ab.py libA.so libB.so ...
Per library: the stage-2 visibility launch (lean engine), the stage-1 march sweep (occupancy engine, 1M rows), the
four geometry chains of a stage-1 step (chain engine) and the bf16 engine; outputs are compared with the first library."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused, ops

dev = torch.device('cuda')
orig = hip._lib


def use(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args

    class Mix(object):
        def __getattr__(self, n):
            return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
    hip._lib = Mix()


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
# ---- visibility net (lean engine) and the bf16 engine
Ns, L = 29487, 104
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64)
ta16, tb16 = ta.to(torch.bfloat16), tb.to(torch.bfloat16)
Q = Ns * L
# ---- occupancy net
Qo = 1 << 20
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
Wo = [torch.randn(o, i, device=dev) * (1.0 / i ** 0.5) for i, o in zip(dims_in, dims_out)]
bo = [torch.randn(o, device=dev) * 0.01 for o in dims_out]
tabo = hip.pe_encode(torch.rand(Qo, 3, device=dev) - 0.5, 6, 64)
# ---- chains
Qc = 262144
params = []
for i, o in zip(dims_in, dims_out):
    params += [(torch.randn(o, i, device=dev) * (1.4 / i ** 0.5)).requires_grad_(), (torch.randn(o, device=dev) * 0.01).requires_grad_()]
pc = (torch.rand(Qc, 3, device=dev) - 0.5).requires_grad_()
names = ['F1 value', 'F2 sweep', 'B1 sweep-adj', 'B2 value-adj']
macs = [39 * 256 + 6 * 65536 + 256 * 217 + 256 * 256 + 2 * 65536 + 256, 7 * 65536 + 256 * 64, 64 * 256 + 7 * 65536 + 64 * 256, 8 * 65536]

ref = {}
best = {}
REPS = int(os.environ.get('AB_REPS', '3'))
for rep in range(REPS):  # libraries interleaved, minimum over the rounds: clocks drift by several % within a process
  for path in sys.argv[1:]:
    fused.TRIM_ACT_KTILES = not path.endswith('+full')  # 'lib.so+full': every layer reads all activation k-tiles
    use(path.replace('+full', ''))
    tag = os.path.basename(path)
    res = {}
    T = best.setdefault(tag, {})

    def rec(key, ms):
        T[key] = min(T.get(key, 1e9), ms)
    packed = fused.pack_relu_mlp(ws, bs, 63, 63, skip_at=3)
    out = torch.empty(Q, 1, device=dev)
    rec('vis lean', timeit(lambda: packed(ta, Q, 1, Ns, tb, Ns, L, out=out)))
    res['vis'] = out.clone()
    p16 = fused.pack_relu_mlp_bf16(ws, bs, 63, 63, 3, hip.OUT_SIGMOID)
    out16 = torch.empty(Q, 1, device=dev)
    rec('vis bf16', timeit(lambda: p16(ta16, Q, 1, Ns, tb16, Ns, L, out=out16), n=10))
    res['vis16'] = out16.clone()
    p16g = fused.pack_relu_mlp_bf16_grouped(ws, bs, 63, 63, 3, hip.OUT_SIGMOID)
    out16g = torch.empty(Q, 1, device=dev)
    rec('vis bf16 grp', timeit(lambda: p16g(ta16, tb, out=out16g), n=10))
    res['vis16g'] = out16g.clone()
    # the V-row backward chain of a stage-2 step: 8 visibility lights x 29487 points, ReLU-mask chain over transposed packs
    Qv = 8 * Ns
    if 'Hv' not in globals():
        globals()['Hv'] = [torch.randn(Qv, 256, device=dev) for _ in range(8)]
        globals()['DZv'] = [torch.empty(Qv, 256, device=dev) for _ in range(8)]
        globals()['gv'] = torch.randn(Qv, 1, device=dev)
    chv = fused.pack_relu_bwd(ws, 3)
    wlast = ws[-1].contiguous()
    rec('V-row bwd', timeit(lambda: chv(None, Qv, a_div=1, a_mod=Qv, rank_init=(gv, wlast), mask=Hv, save=DZv)))
    res['dzv'] = DZv[-1].clone()
    occ = fused.pack_geo_occupancy(Wo, bo, [4], 39)
    oo = torch.empty(Qo, 1, device=dev)
    rec('occ march', timeit(lambda: occ(tabo, Qo, out=oo)))
    res['occ'] = oo.clone()
    chains = fused.pack_geo_chains(params[0::2], params[1::2], [4], 39)
    for it in range(4):
        for q in params: q.grad = None
        hip.PROFILE_EVENTS = []
        logit, feat, grad = ops.GeoFieldFused.apply(pc, 6, 1.0, (4,), True, chains, None, *params)
        (logit.sum() + feat.sum() * 0.1 + (grad * grad).sum()).backward()
        torch.cuda.synchronize()
        ev = hip.PROFILE_EVENTS
        hip.PROFILE_EVENTS = None
        if it > 0:
            for n, (nm, rows, a, b, _f) in zip(names, ev):
                rec(n, a.elapsed_time(b))
            for i_, (nm, rows, a, b, _f) in enumerate(ev[4:]):
                rec('wgrad%d %s' % (i_, nm), a.elapsed_time(b))
    res['logit'], res['feat'], res['grad'] = logit.detach().clone(), feat.detach().clone(), grad.detach().clone()
    res['gW0'], res['gW4'] = params[0].grad.clone(), params[8].grad.clone()
    if not ref:
        ref = res
    elif rep == 0:
        for k in res:
            d = (res[k] - ref[k]).abs().max().item()
            print('   %-22s %-6s max|d| vs first library %.3e (max |ref| %.3e)' % (tag, k, d, ref[k].abs().max().item()))
    hip._lib = orig
flop = {'V-row bwd': 2.0 * 7 * 65536 * 8 * Ns, 'vis lean': 2.0 * 523520 * Q, 'vis bf16': 2.0 * 523520 * Q, 'vis bf16 grp': 2.0 * 523520 * Q, 'occ march': 2.0 * (39 * 256 + 6 * 65536 + 256 * 217 + 256 * 256 + 256) * Qo}
for n, m in zip(names, macs):
    flop[n] = 2.0 * m * Qc
keys = list(next(iter(best.values())).keys())
print('%-14s' % '' + ''.join('%24s' % t[:24] for t in best))
for k in keys:
    print('%-14s' % k[:14] + ''.join('%13.3f ms %6.1f TF' % (best[t][k], flop.get(k, 0) / best[t][k] * 1e-9) for t in best))
print('%-14s' % 'chains total' + ''.join('%13.3f ms          ' % sum(best[t][n] for n in names) for t in best))
