"""Where does the occupancy (sweep) engine lose time?  Same 9-layer 256-wide network, 1M rows:
  A  softplus, input block as k-tiles (the product path)        B  ReLU instead of softplus
  C  ReLU, input block through init tables (like the vis net)    D  softplus, init tables"""
import sys, torch
sys.path.insert(0, '.')
from psnerf_amd import hip, fused
torch.manual_seed(0)
dev = torch.device('cuda')
Q = 1 << 20
d_pe, d_a = 39, 217
dims_in = [39, 256, 256, 256, 256, 256, 256, 256, 256]
dims_out = [256, 256, 256, 217, 256, 256, 256, 256, 257]
W = [torch.randn(o, i, device=dev) * (1.0 / i ** 0.5) for i, o in zip(dims_in, dims_out)]
b = [torch.randn(o, device=dev) * 0.01 for o in dims_out]
tab = hip.pe_encode(torch.rand(Q, 3, device=dev) - 0.5, 6, 64)


def build(act, init):
    layers = []
    n = len(W)
    for li in range(n):
        last = li == n - 1
        a = hip.ACT_NONE if last else act
        Wl, bl = (W[li][:1], b[li][:1]) if last else (W[li], b[li])
        if li == 0:
            layers.append(dict(init_a=Wl, init_b=None, w_act=None, bias=bl, act=a) if init else dict(w_in=Wl, w_act=None, bias=bl, act=a))
        elif li == 4:
            if init:
                layers.append(dict(init_a=Wl[:, d_a:], init_b=None, w_act=Wl[:, :d_a], bias=bl, act=a))
            else:
                layers.append(dict(w_in=Wl[:, d_a:], w_act=Wl[:, :d_a], bias=bl, act=a))
        else:
            layers.append(dict(w_in=None, w_act=Wl, bias=bl, act=a))
    return fused.pack_layers(layers, 2, 0, 1, hip.OUT_OCC, dev)


def timeit(pk, n=5):
    out = torch.empty(Q, 1, device=dev)
    for _ in range(2): pk(tab, Q, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): pk(tab, Q, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

macs = 39 * 256 + 6 * 65536 + 256 * 217 + 256 * 256 + 256
for name, act, init in (('A softplus, k-tiles', hip.ACT_SOFTPLUS100, False), ('B relu, k-tiles', hip.ACT_RELU, False),
                        ('C relu, init tables', hip.ACT_RELU, True), ('D softplus, init tables', hip.ACT_SOFTPLUS100, True)):
    ms = timeit(build(act, init))
    print('%-26s %7.3f ms  %6.1f TF algorithmic' % (name, ms, 2.0 * macs * Q / ms / 1e9))
