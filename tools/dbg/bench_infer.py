import sys, time, torch
sys.path.insert(0, '.')
from psnerf_amd import hip, fused
torch.manual_seed(0)
dev = torch.device('cuda')
Ns, L = 29491, 96
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(4)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(2)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
packed = fused.pack_relu_mlp(ws, bs, 63, 63, skip_at=4)
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64)
Q = Ns * L
out = torch.empty(Q, 1, device=dev)
for _ in range(2):
    packed(ta, Q, 1, Ns, tb, Ns, L, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    packed(ta, Q, 1, Ns, tb, Ns, L, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flops = 2 * 523520 * Q
print('mlp_infer vis: Q=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)' % (Q, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100))
# gemm NT throughput
M = 236000
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
C = torch.empty(M, 256, device=dev)
for name, kw in (('NT bias relu', dict(trans_b=True, bias=b, epi=hip.EPI_BIAS_RELU)), ('NN', dict()),):
    for _ in range(2): hip.gemm(A, W, out=C, **kw)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): hip.gemm(A, W, out=C, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('gemm %s M=%d: %.3f ms %.1f TF' % (name, M, ms, 2 * M * 256 * 256 / ms / 1e9))
dW = torch.empty(256, 256, device=dev)
for sk in (32, 128, 512):
    for _ in range(2): hip.gemm(A, C, trans_a=True, out=dW, split_k=sk)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): hip.gemm(A, C, trans_a=True, out=dW, split_k=sk)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('gemm TN split %d: %.3f ms %.1f TF' % (sk, ms, 2 * M * 256 * 256 / ms / 1e9))
