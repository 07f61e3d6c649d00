"""Time variant builds of csrc/mlp_infer.hip on the stage-2 shading-row launch with split-bf16 weight stages:
    python tools/dbg/bench_lrow_x3_variants.py lib1.so lib2.so ...   (hipcc -shared -fPIC -D... mlp_infer.hip error.hip -o libv.so)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns, L = 29487, 96
dims = [(256, 78)] + [(256, 256)] * 3 + [(256, 256 + 78)] + [(256, 256)] * 2 + [(1, 256)]
Ws = [torch.randn(o, i, device=dev) * (1.4 / i ** 0.5) for o, i in dims]
bs = [torch.randn(o, device=dev) * 0.01 for o, _ in dims]
pe_x, pe_l = torch.randn(Ns, 64, device=dev), torch.randn(L, 64, device=dev)
pk = fused.pack_relu_mlp(Ws, bs, 39, 39, 3, x3=True)
out = torch.empty(L * Ns, 1, device=dev)
orig = hip._lib
f = lambda: pk(pe_x, L * Ns, a_div=1, a_mod=Ns, tab_b=pe_l, b_div=Ns, b_mod=L, out=out)
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    class Mix(object):
        def __getattr__(self, n):
            return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
    hip._lib = Mix()
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print('%-28s %.3f ms' % (os.path.basename(path), e0.elapsed_time(e1) / 5))
    hip._lib = orig
