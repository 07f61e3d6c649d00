"""Time variant builds of csrc/gemm.hip (grouped 256x256 weight gradients): python tools/dbg/bench_tn_variants.py lib1.so ..."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip
dev = torch.device('cuda:0')
Q = 524288
items = [dict(A=torch.randn(Q, 256, device=dev), B=torch.randn(Q, 256, device=dev), A2=torch.randn(Q, 256, device=dev), B2=torch.randn(Q, 256, device=dev), colsum=True) for _ in range(8)]
ref = [c.clone() for c, _ in hip.gemm_tn_grouped(items, 8)]
orig = hip._lib
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    libs.append((os.path.basename(path), lib))
best, errs = {}, {}
for rep in range(3):  # libraries interleaved, minimum over the rounds
    for name, lib in libs:
        class Mix(object):
            def __getattr__(self, n, lib=lib):
                return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
        hip._lib = Mix()
        out = hip.gemm_tn_grouped(items, 8); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): hip.gemm_tn_grouped(items, 8)
        e1.record(); torch.cuda.synchronize()
        best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 5)
        errs[name] = max(float((c - r).abs().max()) for (c, _), r in zip(out, ref))
        hip._lib = orig
for name, _ in libs:
    print('%-30s %.3f ms  %.1f TF  max|d| %.2e' % (name, best[name], 16 * 2 * 256 * 256 * Q / best[name] / 1e9, errs[name]))
