"""Time variant builds of csrc/mlp_infer_bf16.hip: python tools/dbg/bench_bf16_variants.py lib1.so lib2.so ...
(each built with `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -D... mlp_infer_bf16.hip error.hip`)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip, fused

dev = torch.device('cuda')
torch.manual_seed(0)
Ns, L = 29500, 104
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(2)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
packed = fused.pack_relu_mlp_bf16(ws, bs, 63, 63, 3, hip.OUT_SIGMOID)
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64).to(torch.bfloat16)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64).to(torch.bfloat16)
Q = Ns * L
ref = packed(ta, Q, 1, Ns, tb, Ns, L)
flops = 2.0 * (126 * 256 + 5 * 65536 + 382 * 256 + 256) * Q
fns = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.psn_mlp_infer_bf16
    fn.restype = ctypes.c_int
    c = ctypes.c_void_p
    fn.argtypes = [ctypes.POINTER(hip.PsnBf16Desc), c, c, c, ctypes.c_int64, ctypes.c_int64, c, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c, c]
    fns.append((os.path.basename(path), fn))
out = torch.empty(Q, 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
best = {}
for rep in range(4):  # variants interleaved, minimum over the rounds: clocks drift by several % within a process
    for name, fn in fns:
        def run():
            rc = fn(ctypes.byref(packed.desc), packed.w.data_ptr(), packed.final_bias.data_ptr(), ta.data_ptr(), 1, Ns, tb.data_ptr(), Ns, L, Q, out.data_ptr(), st)
            assert rc == 0, rc
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        d = (out - ref).abs().max().item()
        best[name] = min(best.get(name, (1e9, 0))[0], ms), d
for name, _ in fns:
    ms, d = best[name]
    print('%-44s %.3f ms  %.1f TFLOP/s  max|d| vs product build %.2e' % (name, ms, flops / ms * 1e-9, d))
