"""Is the bf16 engine clock / power limited?  The same launch with real data and with all-zero operands (no toggling)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
Ns, L = 29500, 104
Q = Ns * L
flops = 2.0 * 523520 * Q


def run(scale, tag):
    ws = [torch.randn(256, 126, device=dev) * 0.1 * scale] + [torch.randn(256, 256, device=dev) * 0.06 * scale for _ in range(3)] + \
         [torch.randn(256, 382, device=dev) * 0.05 * scale] + [torch.randn(256, 256, device=dev) * 0.06 * scale for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06 * scale]
    bs = [torch.randn(w.shape[0], device=dev) * 0.1 * scale for w in ws]
    ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64) * scale
    tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64) * scale
    for name, pk, a, b in (('bf16', fused.pack_relu_mlp_bf16(ws, bs, 63, 63, 3, hip.OUT_SIGMOID), ta.to(torch.bfloat16), tb.to(torch.bfloat16)),
                           ('fp32', fused.pack_relu_mlp(ws, bs, 63, 63, skip_at=3), ta, tb)):
        out = torch.empty(Q, 1, device=dev)
        best = 1e9
        for rep in range(3):
            for _ in range(3): pk(a, Q, 1, Ns, b, Ns, L, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): pk(a, Q, 1, Ns, b, Ns, L, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print('%-5s %-10s %.3f ms  %.1f TFLOP/s' % (name, tag, best, flops / best * 1e-9), flush=True)


run(1.0, 'real data')
run(0.0, 'all zeros')
run(1.0, 'real data')
