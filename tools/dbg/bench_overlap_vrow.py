"""Would the V-row backward of a stage-2 step hide under the L-row inference launch?  The visibility launch (lean engine, 96 x Ns
rows) and the V-row backward chain (8 x Ns rows, ReLU-mask chain) + its grouped weight gradients, timed alone and on two streams."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns, L, V = 29487, 96, 8
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64)
Q = Ns * L
packed = fused.pack_relu_mlp(ws, bs, 63, 63, skip_at=3)
out = torch.empty(Q, 1, device=dev)
Qv = V * Ns
Hv = [torch.randn(Qv, 256, device=dev) for _ in range(8)]
DZv = [torch.empty(Qv, 256, device=dev) for _ in range(8)]
gv = torch.randn(Qv, 1, device=dev)
chv = fused.pack_relu_bwd(ws, 3)
wlast = ws[-1].contiguous()
items = [dict(A=DZv[j], B=Hv[j], colsum=True) for j in range(7)]

def main_launch():
    packed(ta, Q, 1, Ns, tb, Ns, L, out=out)

def vrow():
    chv(None, Qv, a_div=1, a_mod=Qv, rank_init=(gv, wlast), mask=Hv, save=DZv)
    hip.gemm_tn_grouped(items)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        vrow()
    with torch.cuda.stream(s1):
        main_launch()
    cur.wait_stream(s1); cur.wait_stream(s2)

def both_main_first():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        main_launch()
    with torch.cuda.stream(s2):
        vrow()
    cur.wait_stream(s1); cur.wait_stream(s2)

for rep in range(2):
    a = timeit(main_launch); b = timeit(vrow); c = timeit(both); d = timeit(both_main_first)
    print('main %.3f ms   v-row bwd + wgrad %.3f ms   sum %.3f   two streams (v-row first) %.3f   (main first) %.3f' % (a, b, a + b, c, d))
