#!/usr/bin/env python3
"""Per-rank step time of BASELINE cfg 4 (cfg 3's 32768-pixel batch sharded over N ranks) measured on ONE GPU.

    python tools/strong_projection.py [--out profiles/r04_strong_projection.json] [--steps 50] [--graph]

For px in {32768, 16384, 8192, 4096} (= the rank shard at N = 1, 2, 4, 8) it runs the stage-2 train step of bench.py with the
DATA-PARALLEL code path on -- a process group with backend 'nccl' (= RCCL) in a world of one rank, DataParallel(force=True):
the count all-reduce and the flat-bucket gradient all-reduce are issued every step -- and reports
    wall_ms     wall time per step (K steps, synchronised on both sides),
    host_ms     time the Python loop needs to ISSUE a step (the loop's own duration before the final synchronise; equals
                wall_ms when the step is host-bound),
    kernel_ms   sum of the device kernel durations of one steady-state step (torch.profiler),
    launches    device launches of that step,
and, with --graph, the same step replayed from a HIP graph (psnerf_amd.stage2.graph.GraphedTrainStep).
The projection for N ranks is wall_ms(32768 / N) + the measured all-reduce time of the bucket; speed-up = wall_ms(32768) / that.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r04_strong_projection.json'))
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--pixels', type=int, nargs='*', default=[32768, 16384, 8192, 4096])
    ap.add_argument('--graph', action='store_true')
    ap.add_argument('--queue-ahead', action='store_true', help='also measure gpu_ms: K steps queued behind a spinning kernel')
    ap.add_argument('--spin-cycles', type=float, default=4e8)
    ap.add_argument('--no-profiler', action='store_true')
    ap.add_argument('--no-dp', action='store_true', help='plain single-process step (no process group)')
    args = ap.parse_args()

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import bench
    from psnerf_amd import dist as pdist
    from psnerf_amd.synthetic import stage2_inputs
    import psnerf_amd.stage2 as s2

    if not args.no_dp:
        pdist.init_from_env(backend='nccl', set_device=True, force=True)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    L, V = bench.N_LIGHTS, bench.N_VIS
    l_slt = torch.arange(L, device=dev) + 96 * 3
    res = {'note': __doc__.split('\n')[0], 'steps': args.steps, 'data_parallel_path': not args.no_dp, 'cases': []}

    def make():
        return bench.make_step(dev, dp=None if args.no_dp else pdist.DataParallel(dev, force=True))

    for px in args.pixels:
        inp, gt = stage2_inputs(px, L, V, seed=100, device=dev, with_surface_idx=True)
        ns = int(inp['surface_mask'].sum())
        case = {'pixels': px, 'surface_pixels': ns}
        for mode in (['eager', 'graph', 'graph_single_stream'] if args.graph else ['eager']):
            step = make()
            if mode != 'eager':
                from psnerf_amd.stage2.graph import GraphedTrainStep
                run = GraphedTrainStep(step, overlap_small_nets=None if mode == 'graph' else False, adopt_inputs=True)
                fn = lambda: run.step(inp, gt, l_slt, train_order=False)
            else:
                fn = lambda: step.step(inp, gt, l_slt, train_order=False)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                terms, _ = fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            # gpu_ms: the same K steps queued BEHIND a spinning kernel that holds the GPU while the host issues all of them: what the
            # GPU needs per step when the host is never in the way (HIP events behind the spin kernel / behind the last step)
            gpu_ms = None
            if args.queue_ahead:
                k = min(args.steps, 30)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(int(args.spin_cycles))
                e0.record()
                for _ in range(k):
                    fn()
                e1.record()
                th = time.perf_counter()
                torch.cuda.synchronize()
                gpu_ms = round(e0.elapsed_time(e1) / k, 4)
                # (if the host needed longer than the spin lasted, the figure is an upper bound: say so)
                case.setdefault('spin_drain_wait_ms', {})[mode] = round((time.perf_counter() - th) * 1e3, 2)
            r = {'wall_ms': round((t2 - t0) / args.steps * 1e3, 4), 'host_ms': round((t1 - t0) / args.steps * 1e3, 4), 'gpu_ms': gpu_ms,
                 'loss': float(terms['total'].detach()),
                 'ray_samples_per_s': round(ns * L / ((t2 - t0) / args.steps), 1)}
            try:
                if args.no_profiler:
                    raise RuntimeError('skipped')
                from torch.profiler import profile, ProfilerActivity
                with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                    fn()
                    torch.cuda.synchronize()
                evs = [e for e in prof.events() if getattr(e, 'device_type', None) is not None and 'cuda' in str(e.device_type).lower()]
                r['kernel_ms'] = round(sum(e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total for e in evs) * 1e-3, 4)
                r['launches'] = len(evs)
            except Exception as e:  # noqa: BLE001
                r['profiler_error'] = '%s: %s' % (type(e).__name__, str(e)[:200])
            if not args.no_dp:
                r['allreduce_bytes'] = step.dp.allreduce_bytes
                r['allreduce_ms_world1'] = round(step.dp.time_allreduce(max(1, step.dp.allreduce_bytes // 4)), 4)
            case[mode] = r
            print(px, mode, r, flush=True)
            del step, fn
            torch.cuda.empty_cache()
        res['cases'].append(case)
    base = res['cases'][0]
    for c in res['cases']:
        for mode in ('eager', 'graph', 'graph_single_stream'):
            if mode in c and mode in base:
                c[mode]['speedup_vs_%d' % base['pixels']] = round(base[mode]['wall_ms'] / c[mode]['wall_ms'], 3)
                c[mode]['ideal'] = round(base['pixels'] / c['pixels'], 3)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, 'w'), indent=1)
    print(json.dumps(res))
    if not args.no_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
