"""The 256 x 256-tile weight-gradient products on the exact fp32 kernel (gemm_tn256_grouped_kernel, v_mfma_f32_32x32x2_f32) vs the
split-bf16 experiment (gemm_tn256_x3_grouped_kernel: three bf16 planes per operand, six partial products, fp32 accumulation):
time at the stage-1 operating point (8 products with a second segment, K = 537k rows), accuracy of both against float64."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip
dev = torch.device('cuda:0')
g = torch.Generator(device='cuda').manual_seed(0)

def items(K, n, seg2, M=256, N=256, cs=True):
    out = []
    for _ in range(n):
        it = dict(A=(torch.randn(K, 256, device=dev, generator=g) * 0.3)[:, :M], B=(torch.randn(K, 256, device=dev, generator=g).abs() * 0.1)[:, :N], colsum=cs)
        if seg2:
            it['A2'], it['B2'] = torch.randn(K, 256, device=dev, generator=g)[:, :M], (torch.randn(K, 256, device=dev, generator=g) * 0.05)[:, :N]
        out.append(it)
    return out

def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

rep = {}
# accuracy: K = 40000 (not a multiple of the k-tiles), M = 217 and N = 256 / 200 as well
acc = {}
for (M, N) in ((256, 256), (217, 256), (256, 200)):
    its = items(40000, 2, True, M, N)  # (views of 256-wide tensors: lda = 256, the 256 x 256-tile path)
    ref = [(it['A'].double().t() @ it['B'].double() + it['A2'].double().t() @ it['B2'].double(), it['A'].double().sum(0)) for it in its]
    for name in ('fp32', 'bf16x6', 'bf16x3', 'bf16'):
        with hip.wgrad_precision(name):
            res = hip.gemm_tn_grouped(its)
        e = max(float((c.double() - r).abs().max() / r.abs().max()) for (c, _), (r, _) in zip(res, ref))
        ec = max(float((s.double() - rs).abs().max() / rs.abs().max()) for (_, s), (_, rs) in zip(res, ref))
        acc['%s M=%d N=%d' % (name, M, N)] = {'max_rel_err_vs_float64': e, 'colsum_max_rel_err': ec}
rep['accuracy'] = acc
K = int(sys.argv[1]) if len(sys.argv) > 1 else 537000
its = items(K, 8, True)
flops = 2.0 * K * 256 * 256 * 16
bytes_ = 16 * K * 1024 * 2.0
for name in ('fp32', 'bf16x6', 'bf16x3', 'bf16') * 2:
    with hip.wgrad_precision(name):
        ms = timeit(lambda: hip.gemm_tn_grouped(its))
    r = rep.setdefault(name, {'ms': []})
    r['ms'].append(round(ms, 3))
for name in ('fp32', 'bf16x6', 'bf16x3', 'bf16'):
    ms = min(rep[name]['ms'])
    rep[name].update({'best_ms': ms, 'tflops_fp32_equivalent': round(flops / ms / 1e9, 1), 'operand_TB_per_s': round(bytes_ / ms / 1e9, 2)})
rep['K'] = K
rep['products'] = '8 x (A^T B + A2^T B2), 256 x 256, column sums of A (incl. split-K reduction)'
print(json.dumps(rep))
