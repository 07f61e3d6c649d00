"""bf16 inference engine vs (a) a float64 emulation of the same roundings, (b) the fp32 engine; plus timing."""
import sys, time
import torch
sys.path.insert(0, '.')
from psnerf_amd import hip, fused


def make_net(din_half=63, depth=8, skip_at=3, seed=0, dev='cuda'):
    g = torch.Generator().manual_seed(seed)
    dims_in = [2 * din_half] + [256] * (depth - 1)
    Ws, bs = [], []
    for li in range(depth):
        fan_in = dims_in[li] + (2 * din_half if li - 1 == skip_at else 0)
        out = 256 if li < depth - 1 else 1
        k = 1.0 / fan_in ** 0.5
        Ws.append(((torch.rand(out, fan_in, generator=g) * 2 - 1) * k * 1.7).to(dev))
        bs.append(((torch.rand(out, generator=g) * 2 - 1) * k).to(dev))
    return Ws, bs


def emulate(Ws, bs, xa, xb, skip_at, sigmoid):
    r = lambda t: t.to(torch.bfloat16).double()
    x = torch.cat([r(xa), r(xb)], dim=1)
    h = None
    n = len(Ws)
    for li in range(n - 1):
        W = r(Ws[li])
        bh = bs[li].to(torch.bfloat16).float()
        b = bh.double() + r(bs[li] - bh)
        inp = x if li == 0 else (torch.cat([h, x], dim=1) if li - 1 == skip_at else h)
        h = r(torch.relu(inp @ W.t() + b).float())
    out = h @ r(Ws[-1]).t() + bs[-1].double()
    return torch.sigmoid(out).float() if sigmoid else out.float()


def main():
    dev = 'cuda'
    torch.manual_seed(1)
    Ws, bs = make_net()
    nA, nB = 1000, 7
    pe_a = torch.zeros(nA, 64, device=dev); pe_a[:, :63] = torch.randn(nA, 63, device=dev).clamp(-1, 1)
    pe_b = torch.zeros(nB, 64, device=dev); pe_b[:, :63] = torch.randn(nB, 63, device=dev).clamp(-1, 1)
    for sig in (False, True):
        pk = fused.pack_relu_mlp_bf16(Ws, bs, 63, 63, 3, hip.OUT_SIGMOID if sig else hip.OUT_NONE)
        out = pk(pe_a.to(torch.bfloat16), nA * nB, a_div=1, a_mod=nA, tab_b=pe_b.to(torch.bfloat16), b_div=nA, b_mod=nB)
        torch.cuda.synchronize()
        xa = pe_a[:, :63].tile(nB, 1); xb = pe_b[:, :63].repeat_interleave(nA, dim=0)
        ref = emulate(Ws, bs, xa, xb, 3, sig)
        p32 = fused.pack_relu_mlp(Ws, bs, 63, 63, 3, hip.OUT_SIGMOID if sig else hip.OUT_NONE)
        o32 = p32(pe_a, nA * nB, a_div=1, a_mod=nA, tab_b=pe_b, b_div=nA, b_mod=nB)
        print('sigmoid' if sig else 'logit', 'vs emulation: max |d| %.3e   vs fp32 engine: max |d| %.3e  rms %.3e  (|out| max %.3f)' % (
            (out - ref).abs().max().item(), (out - o32).abs().max().item(), (out - o32).pow(2).mean().sqrt().item(), o32.abs().max().item()))
    # timing at the bench shape
    nA, nB = 29500, 104
    pe_a = torch.zeros(nA, 64, device=dev); pe_a[:, :63] = torch.randn(nA, 63, device=dev).clamp(-1, 1)
    pe_b = torch.zeros(nB, 64, device=dev); pe_b[:, :63] = torch.randn(nB, 63, device=dev).clamp(-1, 1)
    ta, tb = pe_a.to(torch.bfloat16), pe_b.to(torch.bfloat16)
    pk = fused.pack_relu_mlp_bf16(Ws, bs, 63, 63, 3, hip.OUT_SIGMOID)
    out = torch.empty(nA * nB, 1, device=dev)
    for _ in range(3):
        pk(ta, nA * nB, a_div=1, a_mod=nA, tab_b=tb, b_div=nA, b_mod=nB, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pk(ta, nA * nB, a_div=1, a_mod=nA, tab_b=tb, b_div=nA, b_mod=nB, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 2.0 * (126 * 256 + 5 * 65536 + 382 * 256 + 256) * nA * nB
    print('bf16 engine: %d rows  %.3f ms  %.1f TFLOP/s algorithmic' % (nA * nB, ms, flops / ms * 1e-9))


main()
