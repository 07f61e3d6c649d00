"""A/B of gemm_tn256_grouped_kernel between library builds in one process: python tools/dbg/ab_tn256.py libA.so libB.so.
Interleaved rounds, minimum per case; the outputs of every library are compared bit for bit with the first one's."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
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


torch.manual_seed(0)
cases = [(235896, 6, 256, 256), (524288, 16, 256, 256), (532480, 3, 256, 256), (235896, 12, 256, 256), (100001, 5, 256, 256), (300000, 4, 217, 256), (300000, 4, 256, 217)]
data = {}
for K, n, M, N in cases:
    # (narrow operands are column slices of 256-wide buffers, as the dumps of the 217-output layer are: row stride 256)
    data[(K, n, M, N)] = [dict(A=torch.randn(K, 256, device=dev)[:, :M], B=torch.randn(K, 256, device=dev)[:, :N], colsum=True) for _ in range(n)]
libs = sys.argv[1:]
best = {l: {c: 1e9 for c in cases} for l in libs}
ref = {}
for rnd in range(3):
    for l in libs:
        use(l)
        for c in cases:
            items = data[c]
            out = hip.gemm_tn_grouped(items)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                out = hip.gemm_tn_grouped(items)
            e1.record(); torch.cuda.synchronize()
            best[l][c] = min(best[l][c], e0.elapsed_time(e1) / 4)
            flat = [t for o in out for t in (o if isinstance(o, (tuple, list)) else [o]) if torch.is_tensor(t)]
            if c not in ref:
                ref[c] = [t.clone() for t in flat]
            else:
                assert all(torch.equal(a, b) for a, b in zip(flat, ref[c])), ('outputs differ', l, c)
for c in cases:
    K, n, M, N = c
    print('K=%7d items=%2d %dx%d  ' % c + '  '.join('%s %.3f ms %.1f TF' % (os.path.basename(l)[:18], best[l][c], 2.0 * K * M * N * n / best[l][c] / 1e9) for l in libs))
print('outputs bit-identical across libraries')
