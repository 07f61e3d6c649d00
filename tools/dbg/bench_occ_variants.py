"""Variant builds of csrc/mlp_infer.hip on the stage-1 march sweep (occupancy engine, 1M rows): interleaved, minimum of 3."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Qo = 1 << 20
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
Wo = [torch.randn(o, i, device=dev) * (1.0 / i ** 0.5) for i, o in zip(dims_in, dims_out)]
bo = [torch.randn(o, device=dev) * 0.01 for o in dims_out]
tabo = hip.pe_encode(torch.rand(Qo, 3, device=dev) - 0.5, 6, 64)
occ = fused.pack_geo_occupancy(Wo, bo, [4], 39)
orig = hip._lib
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    libs.append((os.path.basename(path), lib))
best, outs = {}, {}
for rep in range(3):
    for name, lib in libs:
        class Mix(object):
            def __getattr__(self, n, lib=lib):
                return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
        hip._lib = Mix()
        oo = torch.empty(Qo, 1, device=dev)
        for _ in range(2): occ(tabo, Qo, out=oo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): occ(tabo, Qo, out=oo)
        e1.record(); torch.cuda.synchronize()
        best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 5)
        outs[name] = oo.clone()
        hip._lib = orig
ref = outs[libs[0][0]]
for name, _ in libs:
    d = (outs[name] - ref)
    print('%-28s %.3f ms   max|d| vs first %.2e  (elements that differ: %d of %d)' % (name, best[name], d.abs().max().item(), int((d != 0).sum()), Qo))
