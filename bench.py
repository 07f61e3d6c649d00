#!/usr/bin/env python3
"""Headline benchmark: stage-2 BEAR train step (BASELINE.json configs[2]) in ray-samples/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of the stage-2 joint BRDF + normal + visibility + light
optimisation (forward, losses, backward, Adam + SparseAdam, scheduler) on a synthetic BEAR-shaped batch
that is already resident in HBM: 32768 pixels per GPU (90 % on the surface), L = 96 shading lights,
V = 8 visibility-supervision lights, phase-2 of the train_fix schedule (all nets + lights trainable).
A ray-sample is one (surface pixel, shading light) pair: Ns * L per step.  N > 1 shards
pixels across ranks (weak scaling: 32768 px per GPU, the reference trains on all ~10^5 in-mask pixels
per step) with one flat-bucket RCCL all-reduce of the gradients per step.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (the fused 256-wide visibility MLP,
MFMA-bound, algorithmic FLOPs = 2 * 523,520 MAC per row) timed with HIP events on the launch stream, and
``cpu_baseline`` = the CPU oracle (oracle/stage2.py, a verified restatement of the reference) timed on
this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PIXELS, N_LIGHTS, N_VIS, N_LIGHTS_TOTAL = 32768, 96, 8, 1920
VIS_MACS = 523520  # visibility_net MACs per row
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def make_step(device, seed=0):
    import torch
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    torch.manual_seed(seed)  # identical random-init weights on every rank
    net = s2.PSNetwork(conf).to(device)
    g = torch.Generator().manual_seed(seed + 1)
    light_init = torch.nn.functional.normalize(torch.randn(N_LIGHTS_TOTAL, 3, generator=g), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    step = s2.TrainStep(net, conf, N_LIGHTS_TOTAL, light_init.to(device), device)
    step.cur_iter = 5001  # phase 2 of train_fix: every net and the lights are trainable
    return step


def _cpu_steps(n_pixels, steps):
    import torch
    from oracle import stage2 as o2
    from psnerf_amd.synthetic import stage2_inputs
    torch.manual_seed(0)
    conf = o2.bear_conf()
    net = o2.PSNetwork(conf)
    g = torch.Generator().manual_seed(1)
    light_init = torch.nn.functional.normalize(torch.randn(N_LIGHTS_TOTAL, 3, generator=g), dim=-1)
    tr = o2.TrainStep(net, conf, N_LIGHTS_TOTAL, light_init)
    tr.cur_iter = 5001
    inp, gt = stage2_inputs(n_pixels, N_LIGHTS, N_VIS, seed=3)
    ns = int(inp['surface_mask'].sum())
    l_slt = torch.arange(N_LIGHTS)
    tr.step(inp, gt, l_slt, train_order=False)  # warm-up
    best = None
    for _ in range(steps):
        t0 = time.time()
        tr.step(inp, gt, l_slt, train_order=False)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    return ns, best


def cpu_baseline(n_pixels=4096, steps=2):
    """The oracle (port of the reference's stage-2 step) on the host: all cores on a bounded sample of the same
    workload, and one thread (what the reference's own trainer pins, stage2/trainer.py:23) on a smaller one."""
    import torch
    threads = torch.get_num_threads()
    ns, dt = _cpu_steps(n_pixels, steps)
    res = {'value': ns * N_LIGHTS / dt, 'unit': 'ray-samples/s', 'cores': threads, 'kind': 'port',
           'sample': 'oracle/stage2.py TrainStep, %d px (%d surface) x L=%d, V=%d, best of %d timed steps after 1 warm-up, '
                     '%.2f s/step' % (n_pixels, ns, N_LIGHTS, N_VIS, steps, dt)}
    torch.set_num_threads(1)
    try:
        ns1, dt1 = _cpu_steps(1024, 2)
    finally:
        torch.set_num_threads(threads)
    res['single_thread'] = {'value': ns1 * N_LIGHTS / dt1, 'cores': 1,
                            'sample': '1024 px (%d surface), best of 2 timed steps after 1 warm-up, %.2f s/step' % (ns1, dt1)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pixels', type=int, default=N_PIXELS, help='pixels per GPU (default: the benchmark config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL; gloo for 1-GPU dry runs)')
    ap.add_argument('--single-device', action='store_true', help='dry run: every rank uses cuda:0')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from psnerf_amd import dist as pdist, hip
    from psnerf_amd.synthetic import stage2_inputs

    rank, local, world = pdist.init_from_env(backend=args.backend, set_device=not args.single_device)
    if args.single_device:
        local = 0
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)' % (args.gpus, world))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    step = make_step(device)
    inp, gt = stage2_inputs(args.pixels, N_LIGHTS, N_VIS, seed=100 + rank, device=device)
    ns_local = int(inp['surface_mask'].sum())
    l_slt = torch.arange(N_LIGHTS, device=device) + 96 * 3  # the 96 lights of one view

    def one_step():
        return step.step(inp, gt, l_slt, train_order=False)

    for _ in range(args.warmup):
        one_step()
    hip.PROFILE_EVENTS = [] if rank == 0 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms, _ = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    events = hip.PROFILE_EVENTS
    hip.PROFILE_EVENTS = None
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    ns = torch.tensor([ns_local], device=device, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ns, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    ns_total = int(ns.item())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()  # every rank leaves the group cleanly before rank 0 prints
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = ns_total * N_LIGHTS / (dt / args.steps)

    # dominant kernel: fused visibility MLP over L*Ns rows (one launch per step)
    # (the largest launch of the step: (L + V) * Ns rows; the smaller launches are the backward chains of the V rows)
    infer = [(r, a.elapsed_time(b)) for (name, r, a, b) in events if name == 'mlp_infer']
    top = max([r for r, _ in infer]) if infer else 0
    durs = [t for r, t in infer if r == top]
    rows = [r for r, t in infer if r == top]
    roofline = None
    if durs:
        avg_ms = sum(durs) / len(durs)
        flops = 2.0 * VIS_MACS * (sum(rows) / len(rows))
        achieved = flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get('mlp_infer_kernel_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'mfma', 'kernel': 'mlp_infer_kernel', 'achieved': round(achieved, 2),
                    'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    'traffic': traffic, 'avg_launch_ms': round(avg_ms, 3), 'launches': len(durs),
                    'share_of_step': round(avg_ms * len(durs) / args.steps / ms_per_step, 3)}
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # the CPU oracle is timed at N = 1 only (other ranks would idle behind it)
        cpu = cpu_baseline()
    line = {
        'metric': 'ray-samples/sec (train step) on BEAR stage2', 'value': round(value, 1), 'unit': 'ray-samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'stage2 BEAR BRDF+light joint opt (BASELINE configs[2]): %d px/GPU (%d surface total), '
                               'L=96 shading lights, V=8 visibility lights, sgbasis RGB 9 lobes, visibility + vis_loss on, '
                               'train_fix phase 2, full step (fwd+loss+bwd+Adam+SparseAdam)' % (args.pixels, ns_total),
                   'pixels_per_gpu': args.pixels, 'surface_pixels_total': ns_total, 'lights': N_LIGHTS,
                   'vis_lights': N_VIS, 'parallelism': 'pixel-dp%d' % world},
        'loss': round(float(terms['total'].detach()), 6),
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
