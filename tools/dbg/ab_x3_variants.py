"""Timing-only builds of gemm_tn256_x3_grouped_kernel (libx3_<variant>.so: a step without its MFMAs / without the split + LDS writes /
without the row requests): which part of a step the kernel waits for.  python tools/dbg/ab_x3_variants.py lib1.so lib2.so ..."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
dev = torch.device('cuda')
orig = hip._lib
def use(path):
    if path == 'product':
        hip._lib = orig
        return
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    class Mix(object):
        def __getattr__(self, n):
            return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
    hip._lib = Mix()
K = 537000
g = torch.Generator(device='cuda').manual_seed(0)
its = [dict(A=torch.randn(K, 256, device=dev, generator=g), B=torch.randn(K, 256, device=dev, generator=g), A2=torch.randn(K, 256, device=dev, generator=g),
            B2=torch.randn(K, 256, device=dev, generator=g), colsum=True) for _ in range(8)]
for l in ['product'] + sys.argv[1:]:
    use(l)
    best = 1e9
    for _ in range(3):
        hip.gemm_tn_grouped(its, x3=True); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): hip.gemm_tn_grouped(its, x3=True)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 3)
    print('%-40s %.3f ms' % (os.path.basename(l), best))
