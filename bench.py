#!/usr/bin/env python3
"""Headline benchmark: stage-2 BEAR train step (BASELINE.json configs[2]) in ray-samples/s.

    python bench.py --gpus N --steps K --warmup W          (stand-alone: for N > 1 it starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of the stage-2 joint BRDF + normal + visibility + light
optimisation (forward, losses, backward, Adam + SparseAdam, scheduler) on a synthetic BEAR-shaped batch
that is already resident in HBM: 32768 pixels per GPU (90 % on the surface), L = 96 shading lights,
V = 8 visibility-supervision lights, phase-2 of the train_fix schedule (all nets + lights trainable).
The batch is what handoff.ViewSampler.batch hands to the trainer: the reference's dictionary plus 'surface_idx', the
index list of the surface pixels that the sampler builds on the host from the stage-1 mask (an input re-encoded, not
a result: the step then contains no host synchronisation, everything else of the step runs inside the timed region).
A ray-sample is one (surface pixel, shading light) pair: Ns * L per step.  N > 1 shards
pixels across ranks with one flat-bucket RCCL all-reduce of the gradients per step.  The headline line is
WEAK scaling (32768 px per GPU; the reference trains on all ~10^5 in-mask pixels of a view per step); the
``strong`` object on the same line times a FIXED global batch of 262144 pixels split N ways.

Extra objects on the JSON line (all measured after the headline's timed region):
  roofline      dominant kernel (fused 256-wide visibility MLP, MFMA-bound), HIP events on the launch stream from a SECOND,
                separately instrumented pass (the headline's timed region records nothing).  ``achieved`` / ``frac`` count the
                flops the kernel's own algorithm needs (2 x 462,848 MAC per row: the two input-block layers are factorised into
                per-point / per-light tables computed by separate small GEMMs), so frac <= 1; ``reference_formulation`` states the
                reference's unfactorised 523,520 MAC per row at the same duration as an equivalent throughput -- work
                the kernel does not execute, hence no fraction of a peak
  reference_dict  the same step fed the reference's dictionary WITHOUT 'surface_idx' (the index list is then rebuilt from
                the mask inside the timed step: one nonzero() = one host synchronisation per step)
  launches_per_step  device kernel launches of one steady-state step (torch.profiler), hand-written HIP vs torch-eager
  bf16x6_experiment  opt-in experiment, NOT the headline: the same step with the gradient-free L x Ns visibility rows on the
                split-bf16 engine (csrc/mlp_infer_x3.hip): f32-class results (same oracle gate) from the bf16 matrix pipe
  cpu_baseline  the CPU oracle (oracle/stage2.py, a verified restatement of the reference) on this host, N = 1 only
  strong        strong-scaling measurement (fixed 262144-pixel global batch: the weak per-GPU load at N = 8)
  strong_cfg4   BASELINE cfg 4 as BASELINE.md / SURVEY 8e define it: cfg 3's 32768-pixel batch sharded by pixels over the N
                ranks (32768 / N px per rank, full light set), timed eagerly and replayed from HIP graphs
                (psnerf_amd/stage2/graph.py); at N = 1 additionally ``per_rank_projection``: the rank shards of N = 2, 4, 8
                (16384 / 8192 / 4096 px) timed on this one GPU with the data-parallel code path ON (RCCL in a world of one rank)
  parity        N = 1: the cpu_baseline's oracle step and the HIP step on the SAME 4096-px inputs, weights and jitter draw:
                max relative error of sg_rgb_values, relative loss error, PSNR(HIP) - PSNR(oracle) of the rendered batch
  allreduce_ms  average time of one gradient all-reduce of the step's bucket size (N > 1)
  stage1        BASELINE configs[1] (stage-1 BEAR train step, 4096 rays x 128 samples, 256 march steps), N = 1 only:
                ms/step, ray-samples/s, rooflines of the chain engine, the weight-gradient GEMM and the composite
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench_legs import (  # noqa: E402,F401  (workload constants, CPU baselines, parity, stage-1 and strong-scaling legs: bench_legs.py)
    BENCH_PY, N_PIXELS, N_LIGHTS, N_VIS, N_LIGHTS_TOTAL, STRONG_PIXELS, VIS_MACS, VIS_MACS_ISSUED,
    PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, make_step, _cpu_steps, parity_check, cpu_baseline, _cpu_all_cores, cpu_worker,
    sampler_in_loop, stage1_measure, _stage1_cfg1, stage1_cpu_baseline, stage1_parity, settle_gc, time_steps,
    cfg4_case, strong_cfg4, guarded, run_cfg4_child, _free_port)


# ----------------------------------------------------------------------------------------------- self-launch
def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes through torch.distributed.run and
    relay their output.  This parent never touches the GPU (no HIP call, no torch.cuda query), and the ranks are new
    child processes -- nothing is exec'ed from a GPU-initialised process."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)  # 5 s of timed steps: long enough for 5-second SMI sampling to see the GPU busy
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--pixels', type=int, default=N_PIXELS, help='pixels per GPU of the weak-scaling headline')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help="what the headline value measures; 'strong' = the fixed %d-pixel global batch split N ways "
                         '(the other mode is always reported as an extra object)' % STRONG_PIXELS)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stage1', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the strong-scaling / all-reduce extras')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL; gloo for 1-GPU dry runs)')
    ap.add_argument('--single-device', action='store_true', help='dry run: every rank uses cuda:0')
    ap.add_argument('--cfg4-child', action='store_true', help=argparse.SUPPRESS)  # internal: the strong_cfg4 object in a process of its own
    ap.add_argument('--cfg4-graph', action='store_true', help='N > 1: also capture / replay HIP graphs in the strong_cfg4 diagnostic (default: its eager step only)')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)  # internal: one worker of the all-core CPU baseline (never touches the GPU)
    ap.add_argument('--no-cpu-all-cores', action='store_true', help='skip the all-core leg of the CPU baseline (P worker processes)')
    args = ap.parse_args()

    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    import torch  # (device_count() does not initialise the GPU on this image: the parent stays exec-safe)
    have = torch.cuda.device_count()
    if not args.single_device and have < args.gpus:
        # fail fast and loudly instead of hanging in a rendezvous that can never complete
        print('bench.py: --gpus %d requested but only %d GPU(s) are visible on this node' % (args.gpus, have), file=sys.stderr, flush=True)
        raise SystemExit(2)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.cfg4_child:
        device = torch.device('cuda', 0)
        torch.cuda.set_device(device)
        print('CFG4_JSON ' + json.dumps(strong_cfg4(device, None, 0, 1)), flush=True)
        return
    wall = {}
    t_mark = [time.perf_counter()]

    def mark(name):   # wall seconds per leg of this run (reported as 'wall_s': what a driver's clock around the command is spent on)
        now = time.perf_counter()
        wall[name] = round(wall.get(name, 0.0) + now - t_mark[0], 2)
        t_mark[0] = now
    cfg4_n1 = None
    if args.gpus == 1 and not args.no_extra and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        # N = 1: the strong_cfg4 object needs a process group of its own (RCCL in a world of one rank) and captures HIP graphs
        # next to RCCL's watchdog thread -- it runs in a CHILD process, started HERE, before this process has touched the GPU
        # (nothing is ever exec'ed from a GPU-initialised process), so that nothing it does can take the headline line down
        cfg4_n1 = run_cfg4_child()
        mark('strong_cfg4 (child process)')

    import torch.distributed as dist
    from psnerf_amd import dist as pdist, hip, ops as _ops
    from psnerf_amd.synthetic import stage2_inputs
    _ops.STRICT = True   # a device tensor that falls off the fused engines raises instead of quietly taking a slower formulation

    rank, local, world = pdist.init_from_env(backend=args.backend, set_device=not args.single_device)
    if args.single_device:
        local = 0
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    step = make_step(device)
    l_slt = torch.arange(N_LIGHTS, device=device) + 96 * 3  # the 96 lights of one view

    def timed(inp, gt, steps, warmup):
        """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks.  Nothing is recorded inside."""
        for _ in range(warmup):
            step.step(inp, gt, l_slt, train_order=False)
        hip.PROFILE_EVENTS = None
        settle_gc()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            terms, _ = step.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        ns = torch.tensor([int(inp['surface_mask'].sum())], device=device, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(ns, op=dist.ReduceOp.SUM)
        return float(t.item()), int(ns.item()), terms

    def instrumented(inp, gt, steps):
        """Per-kernel durations: HIP events around every C-ABI launch on its launch stream (hip._Prof), in a pass of its own.
        EVERY rank runs the steps (they contain the data-parallel collectives); only rank 0 records."""
        hip.PROFILE_EVENTS = ev = ([] if rank == 0 else None)
        for _ in range(steps):
            step.step(inp, gt, l_slt, train_order=False)
        torch.cuda.synchronize()
        hip.PROFILE_EVENTS = None
        return ev or []

    def count_launches(inp, gt):
        """Device kernel launches of ONE steady-state step, split into the hand-written HIP kernels (namespace psn) and the
        rest (torch-eager at::native / foreach kernels, memsets); None when the profiler is unavailable."""
        if rank != 0:  # the step contains collectives: every rank runs it, rank 0 under the profiler
            step.step(inp, gt, l_slt, train_order=False)
            torch.cuda.synchronize()
            return None
        ran = False
        try:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                ran = True
                step.step(inp, gt, l_slt, train_order=False)
                torch.cuda.synchronize()
            names = [e.name for e in prof.events() if getattr(e, 'device_type', None) is not None
                     and 'cuda' in str(e.device_type).lower()]
            if not names:
                return None
            ours = sum(1 for n in names if 'psn::' in n)
            return {'total': len(names), 'hip_hand_written': ours, 'other': len(names) - ours}
        except Exception as e:  # noqa: BLE001
            if not ran:  # the other ranks are inside the step's collectives: join them whatever the profiler did
                step.step(inp, gt, l_slt, train_order=False)
                torch.cuda.synchronize()
            return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}

    def weak_batch():
        return stage2_inputs(args.pixels, N_LIGHTS, N_VIS, seed=100 + rank, device=device, with_surface_idx=True)

    def strong_batch():
        inp, gt = stage2_inputs(STRONG_PIXELS, N_LIGHTS, N_VIS, seed=100, device=device, with_surface_idx=True)  # same global batch on every rank
        return step.dp.shard_stage2(inp, gt) if world > 1 else (inp, gt)

    head_batch, other_batch = (weak_batch, strong_batch) if args.scaling == 'weak' else (strong_batch, weak_batch)
    inp, gt = head_batch()
    mark('imports, model, batch')
    dt, ns_total, terms = timed(inp, gt, args.steps, args.warmup)
    mark('headline (warm-up + timed steps)')
    px_local = inp['uv'].shape[1]
    events = instrumented(inp, gt, min(args.steps, 10))
    launches = count_launches(inp, gt) if not args.no_extra else None
    mark('instrumented pass + launch count')
    ref_dict = None
    if not args.no_extra and world == 1:
        # the drop-in number: the reference's own dictionary, no 'surface_idx' -- PSNetwork.forward then builds the index list
        # from surface_mask itself (nonzero(): a host synchronisation inside the step)
        inp_ref = {k: v for k, v in inp.items() if k != 'surface_idx'}
        k3 = max(3, min(args.steps, 20))
        dt3, ns3, _ = timed(inp_ref, gt, k3, 2)
        ref_dict = {'value': round(ns3 * N_LIGHTS / (dt3 / k3), 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dt3 / k3 * 1e3, 3),
                    'steps': k3, 'warmup': 2, 'batch': "the reference's model_input keys only (stage2/model/renderer.py:110-125); the "
                    "surface index list is built inside the step"}
        del inp_ref
    mark('reference_dict')
    x6 = None
    if not args.no_extra and world == 1:
        # EXPERIMENT, never the headline: the L shading rows (gradient-free: vis.detach(), renderer.py:197) on the split-bf16
        # engine -- fp32 operands as three bf16 planes, six partial products per multiply, fp32 accumulation -- which passes the
        # same elementwise oracle gate as the exact-fp32 path (tests/test_bf16_gpu.py).  Separate object, own dtype label.
        step.model.train_vis_bf16x6 = True
        try:
            k4 = max(3, min(args.steps, 30))
            dt4, ns4, terms4 = timed(inp, gt, k4, 3)
            ev4 = instrumented(inp, gt, 5)
            xk = [(r, a.elapsed_time(b)) for (name, r, a, b, _f) in ev4 if name == 'mlp_infer_x3']
            x6 = {'value': round(ns4 * N_LIGHTS / (dt4 / k4), 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dt4 / k4 * 1e3, 3),
                  'steps': k4, 'warmup': 3, 'loss': round(float(terms4['total'].detach()), 6),
                  'dtype': 'f32 emulated on the bf16 matrix pipe (3 x bf16 split operands, 6 partial products, f32 accumulate)',
                  'scope': 'the L x Ns shading rows of visibility_net only (conf train.vis_bf16x6); V supervised rows, every gradient '
                           'and all other kernels: exact f32 as in the headline'}
            with hip.wgrad_precision('bf16x6'):   # + the visibility net's 256 x 256 weight gradients (V supervised rows) on the split-bf16 kernel
                dt5, ns5, terms5 = timed(inp, gt, k4, 3)
            x6['with_wgrad_bf16x6'] = {'value': round(ns5 * N_LIGHTS / (dt5 / k4), 1), 'ms_per_step': round(dt5 / k4 * 1e3, 3),
                                       'loss': round(float(terms5['total'].detach()), 6),
                                       'scope': 'as above + psn_gemm_tn_grouped_x3 for the 256 x 256 weight-gradient products of visibility_net'}
            from psnerf_amd import ops
            with hip.wgrad_precision('bf16x3'), ops.chain_precision('bf16x3'):
                # + the backward chains of the 256-wide networks (V supervised rows, normal / albedo nets) and the weight gradients
                # as three bf16 partial products: every large MFMA stream of the step except the V rows' forward on the bf16 pipe
                dt6_, ns6_, terms6_ = timed(inp, gt, k4, 3)
            x6['with_chains_wgrad_bf16x3'] = {'value': round(ns6_ * N_LIGHTS / (dt6_ / k4), 1), 'ms_per_step': round(dt6_ / k4 * 1e3, 3),
                                              'loss': round(float(terms6_['total'].detach()), 6),
                                              'scope': "shading rows bf16x6 + ops.chain_precision('bf16x3') + hip.wgrad_precision('bf16x3') "
                                                       '(2 x bf16 split operands, 3 partial products, ~16 significant bits; gates: tests/test_bf16_gpu.py)'}
            step.model.train_vis_bf16x6, step.model.train_vis_bf16x3 = False, True
            try:
                with hip.wgrad_precision('bf16x3'), ops.chain_precision('bf16x3'):
                    # the shading rows through the exact engine's own kernel on split-bf16 weight stages (PSN_W_BF16X2, three products)
                    dt7_, ns7_, terms7_ = timed(inp, gt, k4, 3)
                    ev7 = instrumented(inp, gt, 5)
                lk = [(r, a.elapsed_time(b)) for (name, r, a, b, _f) in ev7 if name == 'mlp_infer' and r >= 1000000]
                x6['all_bf16x3'] = {'value': round(ns7_ * N_LIGHTS / (dt7_ / k4), 1), 'ms_per_step': round(dt7_ / k4 * 1e3, 3),
                                    'loss': round(float(terms7_['total'].detach()), 6),
                                    'dtype': 'f32 emulated on the bf16 matrix pipe: 2 x bf16 split operands, 3 partial products, f32 accumulate (~16 significant bits)',
                                    'scope': "conf train.vis_bf16x3 (shading rows: mlp_infer_kernel on PSN_W_BF16X2 stages) + ops.chain_precision('bf16x3') + "
                                             "hip.wgrad_precision('bf16x3'); the V rows' forward, the 128-wide network, shading, losses, Adam: exact f32.  "
                                             'Gates: tests/test_bf16_gpu.py (the exact path\'s oracle gate; parameter gradients; convergence windows)'}
                if lk:
                    x6['all_bf16x3']['shading_rows_kernel'] = {'rows_per_launch': lk[0][0], 'avg_launch_ms': round(sum(t for _, t in lk) / len(lk), 3),
                                                               'f32_equivalent_tflops': round(2.0 * VIS_MACS * lk[0][0] / (sum(t for _, t in lk) / len(lk) * 1e-3) / 1e12, 1)}
            finally:
                step.model.train_vis_bf16x6, step.model.train_vis_bf16x3 = True, False
            if xk:
                rows_x, ms_x = xk[0][0], sum(t for _, t in xk) / len(xk)
                eq = 2.0 * VIS_MACS * rows_x / (ms_x * 1e-3) / 1e12
                x6['kernel'] = {'name': 'mlp_infer_x3_kernel', 'rows_per_launch': rows_x, 'avg_launch_ms': round(ms_x, 3),
                                'f32_equivalent_tflops': round(eq, 1), 'vs_f32_mfma_peak': round(eq / PEAK_F32_MFMA_TFLOPS, 3),
                                'bf16_tflops_issued': round(6 * eq, 1), 'bf16_peak': 2500.0, 'bf16_frac': round(6 * eq / 2500.0, 3)}
        except Exception as e:  # noqa: BLE001  (an experiment must never cost the headline its line)
            x6 = dict(x6 or {}, error='%s: %s' % (type(e).__name__, str(e)[:200]))
        finally:
            step.model.train_vis_bf16x6 = False
            step.model.train_vis_bf16x3 = False
    del inp, gt
    mark('bf16 experiments')
    ms_per_step = dt / args.steps * 1e3
    value = ns_total * N_LIGHTS / (dt / args.steps)
    in_loop = None
    if not args.no_extra and world == 1:
        try:
            in_loop = sampler_in_loop(device, step)
            in_loop['value_over_headline'] = round(in_loop['value'] / value, 4)
        except Exception as e:  # noqa: BLE001
            in_loop = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}

    mark('sampler_in_loop')
    other = None
    allreduce_ms = None
    if not args.no_extra:
        inp, gt = other_batch()
        k2 = max(3, min(args.steps, 10 if args.scaling == 'weak' else 20))
        dt2, ns2, _ = timed(inp, gt, k2, 2)
        other = {'scaling': 'strong' if args.scaling == 'weak' else 'weak', 'value': round(ns2 * N_LIGHTS / (dt2 / k2), 1),
                 'unit': 'ray-samples/s', 'ms_per_step': round(dt2 / k2 * 1e3, 3), 'steps': k2, 'warmup': 2,
                 'pixels_per_gpu': inp['uv'].shape[1], 'surface_pixels_total': ns2,
                 'global_pixels': STRONG_PIXELS if args.scaling == 'weak' else args.pixels * world}
        del inp, gt
        if world > 1:
            allreduce_ms = round(step.dp.time_allreduce(step.dp.allreduce_bytes // 4), 4)
    bucket_bytes = step.dp.allreduce_bytes
    cfg4 = cfg4_n1

    mark('other scaling mode')
    stage1 = None
    if world == 1 and not args.no_stage1:
        del step
        torch.cuda.empty_cache()
        stage1 = stage1_measure(device)
        if not args.no_cpu_baseline:  # BASELINE configs[0] = the stage-1 CPU case: oracle timing + one-step parity at 512 rays x 64 samples
            for key, fn in (('parity', lambda: stage1_parity(device)), ('cpu_baseline', stage1_cpu_baseline)):
                try:
                    stage1[key] = fn()
                except Exception as e:  # noqa: BLE001
                    stage1[key] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    mark('stage1 (step, parity, CPU baseline)')
    line = None
    if rank == 0:
        # dominant kernel: fused visibility MLP over (L + V) * Ns rows, one launch per step (the smaller launches of the
        # same engine are the 128- / 64-wide BRDF / normal nets)
        infer = [(r, a.elapsed_time(b)) for (name, r, a, b, _f) in events if name == 'mlp_infer']
        top = max([r for r, _ in infer]) if infer else 0
        durs = [t for r, t in infer if r == top]
        roofline = None
        if durs:
            avg_ms = sum(durs) / len(durs)
            algo = 2.0 * VIS_MACS * top / (avg_ms * 1e-3) / 1e12       # the reference's unfactorised formulation
            issued = 2.0 * VIS_MACS_ISSUED * top / (avg_ms * 1e-3) / 1e12  # the flops of the algorithm this kernel runs
            traffic = comp = busy = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(pmc) and px_local == N_PIXELS:
                try:
                    j = json.load(open(pmc))
                    traffic, comp, busy = j.get('hbm_side_bytes_per_launch'), j.get('compulsory_bytes_per_launch'), j.get('mfma_busy_frac')
                except Exception:
                    traffic = None
            roofline = {'bound': 'mfma', 'kernel': 'mlp_infer_kernel<false,16,0>', 'achieved': round(issued, 2),
                        'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(issued / PEAK_F32_MFMA_TFLOPS, 4),
                        'flops_per_row': 2 * VIS_MACS_ISSUED, 'rows_per_launch': top, 'avg_launch_ms': round(avg_ms, 3),
                        'launches': len(durs), 'share_of_step': round(avg_ms / ms_per_step, 3),
                        'reference_formulation': {'flops_per_row': 2 * VIS_MACS, 'equivalent_tflops': round(algo, 2),
                                                  'note': "the reference's unfactorised 523,520 MAC per row at this launch's duration: a "
                                                          'throughput EQUIVALENT (work the kernel does not execute), not a roofline fraction'},
                        'traffic': traffic, 'algorithmic_bytes': comp,
                        'traffic_over_algorithmic': round(traffic / comp, 2) if (traffic and comp) else None,
                        'pmc_mfma_busy_frac': busy,
                        'note': 'achieved = 2 x 462,848 MAC per row x rows / avg launch duration (HIP events, second pass): the two layers '
                                'that read [pe(x) | pe(l)] start from per-point / per-light tables, W [pe(x) | pe(l)] = W_a pe(x) + W_b pe(l), '
                                'computed once per point / light by separate small GEMMs (DESIGN.md 3).  traffic = 2 x FETCH_SIZE + WRITE_SIZE of the '
                                'dispatch (rocprofv3 PMC, profiles/pmc_traffic.json; Infinity-Cache hits included), algorithmic_bytes = '
                                'compulsory HBM bytes (tables and weights once, outputs, activation dumps of the V supervised rows).  '
                                'Round 6: the workgroups visit the (light, point) row set point-tile-major (psn_mlp_block_order), so the '
                                'per-point init table is read from the fabric once instead of once per light'}
        cpu = parity = None
        if not args.no_cpu_baseline and world == 1:  # the CPU oracle is timed at N = 1 only (other ranks would idle behind it)
            try:
                parity = parity_check(device)
            except Exception as e:  # noqa: BLE001
                parity = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
            cpu = cpu_baseline(all_core_workers=not args.no_cpu_all_cores)
        mark('stage2 parity + CPU baseline')
        line = {
            'metric': 'ray-samples/sec (train step) on BEAR stage2', 'value': round(value, 1), 'unit': 'ray-samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'stage2 BEAR BRDF+light joint opt (BASELINE configs[2]): %d px/GPU (%d surface total), '
                                   'L=96 shading lights, V=8 visibility lights, sgbasis RGB 9 lobes, visibility + vis_loss on, '
                                   'train_fix phase 2, full step (fwd+loss+bwd+Adam+SparseAdam); the batch carries its V supervision '
                                   'lights ready-made -- the per-step train.vis_plus draw of bear.conf:29 (stage2/trainer.py:384-392: one '
                                   'np.random.choice + a two-index device gather, psnerf_amd.stage2.trainer.VisPlus) is NOT inside the timed '
                                   'step, as BASELINE cfg 3 does not list it' % (px_local, ns_total),
                       'pixels_per_gpu': px_local, 'surface_pixels_total': ns_total, 'lights': N_LIGHTS,
                       'vis_lights': N_VIS, 'parallelism': 'pixel-dp%d' % world,
                       'batch': "ViewSampler.batch layout: reference dictionary + 'surface_idx' (host-built index list of the surface mask)"},
            'loss': round(float(terms['total'].detach()), 6),
            'roofline': roofline, 'cpu_baseline': cpu,
            'cpu_baseline_single_thread': cpu.get('single_thread') if cpu else None,   # (what the reference's trainer pins, stage2/trainer.py:23)
            'cpu_baseline_all_cores': cpu.get('all_cores') if cpu else None,
            'reference_dict': ref_dict, 'launches_per_step': launches,
            'bf16x6_experiment': x6, 'sampler_in_loop': in_loop,
            ('strong' if args.scaling == 'weak' else 'weak'): other,
            'strong_cfg4': cfg4, 'parity': parity,
            'allreduce_ms': allreduce_ms, 'allreduce_bytes': bucket_bytes if world > 1 else None,
            'stage1': stage1,
            'wall_s': wall,
        }
    if not args.no_extra and world > 1:
        # The diagnostic object runs LAST, behind a watchdog: it captures HIP graphs around RCCL collectives on every rank, a path no
        # multi-GPU node has exercised yet -- if it does not come back, rank 0 still prints the headline line (assembled above).
        cfg4 = guarded(lambda: strong_cfg4(device, step.dp, rank, world, graph_at_n=args.cfg4_graph or os.environ.get('PSN_BENCH_CFG4_GRAPH') == '1'), line, rank)
        if line is not None:
            line['strong_cfg4'] = cfg4
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()  # every rank leaves the group cleanly before rank 0 prints
    if rank != 0:
        return
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
