#!/usr/bin/env python3
"""Opt-in variant of the headline step, measured for information only (NOT the headline: the reference computes in
fp32 and bench.py never enables this): `train.vis_bf16` evaluates the L shading-light visibility rows -- which enter
the loss detached (stage2/model/renderer.py:197) -- on the bf16 MFMA engine; everything that receives a gradient stays
in fp32.  Same configuration, inputs and weights as bench.py; prints one JSON line with both step times and the
deviation of the first step's loss terms and gradients from the fp32 step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from psnerf_amd.synthetic import stage2_inputs
    dev = torch.device('cuda:0')
    inp, gt = stage2_inputs(bench.N_PIXELS, bench.N_LIGHTS, bench.N_VIS, seed=100, device=dev)
    ns = int(inp['surface_mask'].sum())
    l_slt = torch.arange(bench.N_LIGHTS, device=dev) + 96 * 3
    res = {}
    first = {}
    for mode in ('fp32', 'bf16'):
        step = bench.make_step(dev)
        step.model.train_vis_bf16 = mode == 'bf16'
        torch.manual_seed(7)
        terms, _ = step.step(inp, gt, l_slt, train_order=False)
        first[mode] = ({k: float(v.detach()) for k, v in terms.items()},
                       torch.cat([p.grad.flatten() for p in step.model.parameters() if p.grad is not None and not p.grad.is_sparse]).clone())
        for _ in range(2):
            step.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[mode] = {'ms_per_step': dt * 1e3, 'value': ns * bench.N_LIGHTS / dt}
    (t32, g32), (t16, g16) = first['fp32'], first['bf16']
    out = {'metric': 'ray-samples/sec (train step) on BEAR stage2, opt-in train.vis_bf16 (information only)', 'unit': 'ray-samples/s',
           'fp32': res['fp32'], 'vis_bf16': res['bf16'], 'speedup': res['fp32']['ms_per_step'] / res['bf16']['ms_per_step'],
           'first_step_loss_terms_fp32': t32, 'first_step_loss_terms_vis_bf16': t16,
           'first_step_grad_rel_l2_diff': float((g16 - g32).norm() / g32.norm())}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
