"""gemm_tn256_grouped_kernel (exact fp32): time per launch against the number of 256 x 256 products and K -- where does the rate at the
stage-2 operating point (6 products, K = 235,896: 0.71 of the peak) go?   python tools/dbg/bench_tn256_sweep.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import hip
dev = torch.device('cuda:0')
g = torch.Generator(device='cuda').manual_seed(0)


def items(K, n, cs=True):
    return [dict(A=torch.randn(K, 256, device=dev, generator=g) * 0.3, B=torch.randn(K, 256, device=dev, generator=g).abs() * 0.1, colsum=cs) for _ in range(n)]


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


rows = []
for K in (58974, 117948, 235896, 471792):
    for n in (2, 3, 6, 12):
        if K * n > 471792 * 6:
            continue
        its = items(K, n)
        ms = timeit(lambda: hip.gemm_tn_grouped(its))
        ms_nocs = timeit(lambda: hip.gemm_tn_grouped([dict(A=i['A'], B=i['B']) for i in its]))
        tf = 2.0 * K * 65536 * n / ms / 1e9
        rows.append({'K': K, 'products': n, 'ms': round(ms, 4), 'ms_no_colsum': round(ms_nocs, 4), 'tflops': round(tf, 1), 'ideal_ms_at_157': round(2.0 * K * 65536 * n / 157.3e9, 4)})
        print(json.dumps(rows[-1]), flush=True)
        del its
