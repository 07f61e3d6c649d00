#!/usr/bin/env python3
"""HBM roofline of the alpha-composite kernels (forward 20 S + 16 B per ray, backward 36 S + 16 B per ray;
the shadow-ray variant of stage1/model/rendering.py:405-406 reads alpha only: 4 S + 4 B per ray).  Rays x samples
sized so that the working set is far beyond the 256 MB Infinity Cache.  Prints one JSON line; under
`rocprofv3 --kernel-trace --stats` the same command gives the per-kernel averages committed in profiles/."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=2 * 1024 * 1024)
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    import torch
    from psnerf_amd import hip
    dev = torch.device('cuda:0')
    N, S = args.rays, args.samples
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.rand(N, S, device=dev, generator=g) * 0.1
    rgb = torch.rand(N, S, 3, device=dev, generator=g)
    d_rgb = torch.rand(N, 3, device=dev, generator=g)
    d_acc = torch.rand(N, device=dev, generator=g)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    res = {}
    cases = [
        ('fwd', lambda: hip.composite_fwd(alpha, rgb, True), (20 * S + 16) * N),
        ('fwd_no_weights', lambda: hip.composite_fwd(alpha, rgb, True, need_weights=False), (16 * S + 16) * N),
        ('bwd', lambda: hip.composite_bwd(alpha, rgb, d_rgb, d_acc, True), (36 * S + 16) * N),
        ('fwd_acc_only', lambda: hip.composite_fwd(alpha, None, False, need_weights=False), (4 * S + 4) * N),
    ]
    for name, fn, nbytes in cases:
        ms = timed(fn)
        gbs = nbytes / ms * 1e-6
        res[name] = {'ms': ms, 'algorithmic_bytes': nbytes, 'achieved_GBps': gbs, 'frac_of_peak': gbs / PEAK_HBM_GBS}
    print(json.dumps({'metric': 'alpha composite, achieved HBM GB/s (algorithmic bytes / kernel time)', 'rays': N, 'samples': S,
                      'peak_GBps': PEAK_HBM_GBS, 'kernels': res}))


if __name__ == '__main__':
    main()
