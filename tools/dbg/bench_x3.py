"""Visibility network (bear.conf shape) on the L x Ns light-major rows of the bench batch: exact-fp32 engine vs the bf16 engine
vs the split-bf16 ("bf16x6") engine: ms per launch, reference-network TFLOP/s, max error against float64 on a row sample."""
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused, ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
Ns, L = int(os.environ.get('NS', 29487)), int(os.environ.get('L', 96))
dims = [(256, 126)] + [(256, 256)] * 3 + [(256, 382)] + [(256, 256)] * 3 + [(1, 256)]
Ws = [(torch.rand(o, i, device=dev) * 2 - 1) * (1.7 / i ** 0.5) for o, i in dims]
bs = [(torch.rand(o, device=dev) * 2 - 1) * (1.0 / i ** 0.5) for o, i in dims]
xa = hip.pe_encode(torch.rand(Ns, 3, device=dev) * 1.2 - 0.6, 10, 64)
xb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64)
if os.environ.get('ZERO') == '1':  # power test: no toggling in the multipliers
    Ws = [w * 0 for w in Ws]; bs = [b * 0 for b in bs]; xa = xa * 0; xb = xb * 0
macs = sum(o * i for o, i in dims)
rows = Ns * L


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, out


p32 = fused.pack_relu_mlp(Ws, bs, 63, 63, 3)
p16 = fused.pack_relu_mlp_bf16_grouped(Ws, bs, 63, 63, 3)
px3 = fused.pack_relu_mlp_x3_grouped(Ws, bs, 63, 63, 3)
xa16 = xa.to(torch.bfloat16)
res = {}
t32, o32 = timeit(lambda: p32(xa, rows, a_div=1, a_mod=Ns, tab_b=xb, b_div=Ns, b_mod=L))
t16, o16 = timeit(lambda: p16(xa16, xb))
tx3, ox3 = timeit(lambda: px3(xa, xb))
# float64 reference on a sample of rows
g = torch.Generator().manual_seed(1)
sel_n = torch.randint(0, Ns, (2048,), generator=g).to(dev)
sel_l = torch.randint(0, L, (2048,), generator=g).to(dev)
x = torch.cat([xa[sel_n, :63], xb[sel_l, :63]], dim=1).double()
h = None
for li in range(8):
    inp = x if li == 0 else (torch.cat([h, x], dim=1) if li - 1 == 3 else h)
    h = torch.relu(inp @ Ws[li].double().t() + bs[li].double())
ref = (h @ Ws[8].double().t() + bs[8].double())[:, 0]
ridx = sel_l * Ns + sel_n
for name, t, o in (('fp32', t32, o32), ('bf16', t16, o16), ('bf16x6', tx3, ox3)):
    err = float((o.reshape(-1)[ridx].double() - ref).abs().max())
    res[name] = {'ms': round(t, 3), 'tflops_reference_network': round(2.0 * macs * rows / t / 1e9, 1), 'max_abs_err_vs_f64': err}
    print(name, res[name], flush=True)
res['rows'] = rows
res['output_scale'] = float(ref.abs().max())
print(json.dumps(res))
