#!/usr/bin/env python3
"""Shadow-ray light visibility of shape_extract (stage1/model/rendering.py:378-408): Ns surface points x 96 lights x 128
samples.  'dense' = every sample through the occupancy network (what the reference does, and round 1 did); 'in-box' =
only the samples inside the +-1.1 cube (the others are zeroed by the reference after the fact).  Same results, bit for bit."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--lights', type=int, default=96)
    args = ap.parse_args()
    import torch
    from psnerf_amd import hip
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    from psnerf_amd.synthetic import stage1_cfg
    dev = torch.device('cuda:0')
    cfg = stage1_cfg('bear')
    torch.manual_seed(42)
    net = NeuralNetwork(cfg)
    ren = Renderer(net, cfg, device=dev)
    g = torch.Generator().manual_seed(0)
    surf = (torch.nn.functional.normalize(torch.randn(args.points, 3, generator=g), dim=-1) * 0.6).to(dev)  # the geometric-init sphere
    ld = torch.nn.functional.normalize(torch.randn(args.lights, 3, generator=g), dim=-1).to(dev)
    S = 128

    def dense():
        t = torch.linspace(0, 1, steps=S, device=dev).view(1, S, 1)
        d = 0.1 * (1.0 - t) + 3.5 * t
        outs = []
        per = max(1, (1 << 22) // (args.points * S))
        for l0 in range(0, args.lights, per):
            p = surf[None, :, None, :] + ld[l0:l0 + per, None, None, :] * d[None]
            alpha = ren._occ(p.reshape(-1, 3)).view(-1, S)
            inside = torch.logical_and((p <= 1.1).all(dim=-1), (p >= -1.1).all(dim=-1)).view(-1, S)
            alpha = torch.where(inside, alpha, torch.zeros_like(alpha)).contiguous()
            outs.append(1 - hip.composite_fwd(alpha, None, False, need_weights=False)[2])
        return torch.cat(outs)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    with torch.no_grad():
        v_new, t_new = timed(lambda: ren.light_visibility(surf=surf, light_dir=ld))
        n_in, n_all = ren.last_shadow_stats if args.lights * args.points * S <= (1 << 24) else (None, None)
        v_old, t_old = timed(dense)
        # opt-in split-bf16 ("bf16x6") occupancy engine: fp32-class arithmetic on the bf16 matrix pipe (csrc/mlp_infer_x3.hip)
        net.inference_precision = 'bf16x6'
        v_x3, t_x3 = timed(lambda: ren.light_visibility(surf=surf, light_dir=ld))
        net.inference_precision = 'fp32'
    rows = args.points * args.lights * S
    print(json.dumps({'workload': 'light_visibility: %d surface points x %d lights x %d samples = %.3g rows' % (args.points, args.lights, S, rows),
                      'dense_seconds': round(t_old, 4), 'in_box_seconds': round(t_new, 4), 'speedup': round(t_old / t_new, 2),
                      'max_abs_diff': float((v_new - v_old).abs().max()), 'bit_identical': bool(torch.equal(v_new, v_old)),
                      'mean_visibility': float(v_new.mean()),
                      'bf16x6_seconds': round(t_x3, 4), 'bf16x6_speedup_vs_in_box': round(t_new / t_x3, 3),
                      'bf16x6_max_abs_diff_vs_fp32': float((v_x3 - v_new).abs().max()),
                      'bf16x6_note': 'opt-in (NeuralNetwork.inference_precision = "bf16x6"): same compaction, occupancy network on the '
                                     'split-bf16 engine; dtype label: f32 emulated on the bf16 matrix pipe, never the headline'}))


if __name__ == '__main__':
    main()
