"""Time variant builds of csrc/mlp_infer.hip on the visibility-net launch: python tools/dbg/bench_infer_variants.py lib1.so ..."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns, L = 29487, 104
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
packed = fused.pack_relu_mlp(ws, bs, 63, 63, skip_at=3)
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64)
Q = Ns * L
out = torch.empty(Q, 1, device=dev)
orig = hip._lib
flops = 2.0 * 523520 * Q
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    class Mix(object):
        def __getattr__(self, n):
            return getattr(lib, n) if hasattr(lib, n) else getattr(orig, n)
    hip._lib = Mix()
    for _ in range(2): packed(ta, Q, 1, Ns, tb, Ns, L, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): packed(ta, Q, 1, Ns, tb, Ns, L, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%-28s %.3f ms  %.1f TFLOP/s' % (os.path.basename(path), ms, flops / ms * 1e-9))
    hip._lib = orig
