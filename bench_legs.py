"""The legs of bench.py other than the headline's timed loop: workload constants, the step factory, the CPU baselines (oracle/, timed on the
host cores), one-step parity, the data pipeline in the loop, the stage-1 measurement (BASELINE configs[1] and [0]) and the strong-scaling
diagnostic of BASELINE cfg 4.  bench.py imports everything here and holds the contract: argument parsing, the timed region, the JSON line."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
BENCH_PY = os.path.join(ROOT, 'bench.py')  # the entry point the child processes (--cpu-worker, --cfg4-child) are started through

N_PIXELS, N_LIGHTS, N_VIS, N_LIGHTS_TOTAL = 32768, 96, 8, 1920
STRONG_PIXELS = 8 * N_PIXELS  # fixed global batch of the strong-scaling line (= the weak batch of 8 GPUs)
VIS_MACS = 523520  # visibility_net MACs per row
VIS_MACS_ISSUED = 462848  # MACs the kernel issues per row: 523,520 - 2 x 126 x 256 (init tables) + the final layer padded to 16 outputs
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def make_step(device, seed=0, dp=None):
    import torch
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    torch.manual_seed(seed)  # identical random-init weights on every rank
    net = s2.PSNetwork(conf).to(device)
    g = torch.Generator().manual_seed(seed + 1)
    light_init = torch.nn.functional.normalize(torch.randn(N_LIGHTS_TOTAL, 3, generator=g), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    step = s2.TrainStep(net, conf, N_LIGHTS_TOTAL, light_init.to(device), device, dp=dp)
    step.cur_iter = 5001  # phase 2 of train_fix: every net and the lights are trainable
    return step


# ----------------------------------------------------------------------------------------------- CPU baseline
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


def parity_check(device, n_pixels=4096):
    """The metric's "PSNR parity" on the bench line: one full train step of the oracle (CPU) and of the HIP path on the same
    4096-px x L = 96 x V = 8 batch (the cpu_baseline's sample), the same weights and the same injected jitter draw."""
    import math
    import torch
    from oracle import stage2 as o2  # checker only
    import psnerf_amd.stage2 as s2
    from psnerf_amd.synthetic import stage2_inputs
    torch.manual_seed(0)
    onet = o2.PSNetwork(o2.bear_conf())
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(onet.state_dict())
    net.to(device)
    g = torch.Generator().manual_seed(1)
    light_init = torch.nn.functional.normalize(torch.randn(N_LIGHTS_TOTAL, 3, generator=g), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    ostep = o2.TrainStep(onet, o2.bear_conf(), N_LIGHTS_TOTAL, light_init)
    step = s2.TrainStep(net, s2.bear_conf(), N_LIGHTS_TOTAL, light_init.to(device), device)
    ostep.cur_iter = step.cur_iter = 5001
    inp, gt = stage2_inputs(n_pixels, N_LIGHTS, N_VIS, seed=3)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=g) * 0.01
    l_slt = torch.arange(N_LIGHTS) + 96 * 3
    ot, oo = ostep.step(inp, gt, l_slt, train_order=False, noise={'xyz': nz})
    pt, po = step.step({k: v.to(device) for k, v in inp.items()}, {k: v.to(device) for k, v in gt.items()}, l_slt.to(device),
                       train_order=False, noise={'xyz': nz.to(device)})
    a, b = po['sg_rgb_values'].detach().cpu().double(), oo['sg_rgb_values'].detach().double()
    m = (inp['surface_mask'] & inp['object_mask']).expand(N_LIGHTS, -1)

    def psnr(x):  # stage2/trainer.py:268-276 on the masked pixels of all lights
        return -10.0 * math.log10(float(((x[m] - gt['rgb'].double()[m]) ** 2).mean()))
    # elementwise bound of the parity tests: 1e-4 |ref| + 1e-6 (colours in [0, 1])
    ratio = ((a - b).abs() / (1e-4 * b.abs() + 1e-6))
    lo, lh = float(ot['total'].detach()), float(pt['total'].detach())
    pd = max(float((p.detach().cpu() - q.detach()).abs().max()) for p, q in zip(net.parameters(), onet.parameters()))
    return {'sample': '%d px (%d surface) x L=%d, V=%d, phase 2, one full step, same weights / batch / jitter draw' % (n_pixels, ns, N_LIGHTS, N_VIS),
            'sg_rgb_max_rel_err': float(((a - b).abs().max() / b.abs().max())), 'sg_rgb_worst_over_bound': round(float(ratio.max()), 3),
            'sg_rgb_frac_beyond_bound': float((ratio > 1).double().mean()), 'bound': '1e-4 |ref| + 1e-6 elementwise',
            'loss_hip': lh, 'loss_oracle': lo, 'loss_rel_err': abs(lh - lo) / abs(lo),
            'psnr_hip_db': round(psnr(a), 5), 'psnr_oracle_db': round(psnr(b), 5), 'psnr_diff_db': round(psnr(a) - psnr(b), 6),
            'max_param_diff_after_step': pd,
            'horizon': 'one step here; 300 stage-2 / 200 stage-1 steps: tests/test_convergence_gpu.py (PSNR within 0.05 dB)'}


def cpu_baseline(n_pixels=4096, steps=3, all_core_workers=True):
    """The oracle (port of the reference's stage-2 step) on the host cores.  The thread count is swept on a 1024-pixel
    sample (1 warm-up + 2 timed steps each) over {1, 8, 16, 32} (more threads than that only oversubscribe the eager
    CPU kernels: 256 threads measured 1.1 k ray-samples/s against 203 k with 16); the best count is then timed on a bounded
    4096-pixel sample of the same workload (min of 3 after 1 warm-up).  One thread is what the reference's own trainer
    pins (stage2/trainer.py:23) and is reported beside it."""
    import torch
    nproc = os.cpu_count() or 1
    t_all = torch.get_num_threads()
    sweep = {}
    try:
        for th in sorted({t for t in (1, 16, 32) if t <= nproc} | {min(8, nproc)}):   # (rounds 1-5 swept 1 / 8 / 16 / 32 / 64: the best was 16 on every 256-core box)
            torch.set_num_threads(th)
            ns1, dt1 = _cpu_steps(1024, 2)
            sweep[th] = ns1 * N_LIGHTS / dt1
        best_th = max(sweep, key=sweep.get)
        torch.set_num_threads(best_th)
        ns, dt = _cpu_steps(n_pixels, steps)
        # BASELINE.md section 3 also names "all cores".  Every host core as a torch thread of ONE process only oversubscribes the
        # eager CPU kernels of this size (round 5, 256-core host: 174 ray-samples/s at 256 threads against 243 k at 16), so the
        # all-core figure is taken the way a host would actually be filled: P = cores / best_th independent worker processes of
        # best_th threads each, every one running the same oracle step on its own 1024-pixel batch at the same time (pixel data
        # parallelism without the gradient exchange -- an upper bound for the host); measured in THIS run, summed over the workers
        all_cores = None
        if all_core_workers:
            try:
                all_cores = _cpu_all_cores(best_th, nproc)
            except Exception as e:  # noqa: BLE001
                all_cores = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    finally:
        torch.set_num_threads(t_all)
    return {'value': ns * N_LIGHTS / dt, 'unit': 'ray-samples/s', 'cores': best_th, 'kind': 'port', 'all_cores': all_cores,
            'sample': 'oracle/stage2.py TrainStep, %d px (%d surface) x L=%d, V=%d, min of %d timed steps after 1 warm-up, '
                      '%.2f s/step, %d threads = the best of the sweep' % (n_pixels, ns, N_LIGHTS, N_VIS, steps, dt, best_th),
            'host_cores': nproc,
            'thread_sweep_1024px': {str(k): round(v, 1) for k, v in sorted(sweep.items())},
            'single_thread': {'value': sweep.get(1), 'cores': 1, 'sample': '1024 px, min of 2 timed steps after 1 warm-up'}}


def _cpu_all_cores(threads, nproc, n_pixels=512, steps=1, max_workers=16):
    """P worker processes x ``threads`` torch threads, all timing ``_cpu_steps(n_pixels, steps)`` concurrently (children of this
    process that never touch the GPU: `bench.py --cpu-worker`); value = the sum of the workers' rates."""
    import subprocess
    P = max(1, min(max_workers, nproc // max(threads, 1)))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    cmd = [sys.executable, BENCH_PY, '--cpu-worker', '%d,%d,%d' % (n_pixels, steps, threads)]
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
    rates = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
            rates.append(float(json.loads(out.strip().splitlines()[-1])['rate']))
        except Exception:  # noqa: BLE001  (a worker that failed or timed out contributes nothing)
            pr.kill()
    return {'value': sum(rates), 'unit': 'ray-samples/s', 'cores': P * threads, 'workers': len(rates),
            'sample': '%d concurrent worker processes x %d threads, each oracle/stage2.py TrainStep on its own %d-px batch (min of %d timed '
                      'steps after 1 warm-up); the sum of their rates, no gradient exchange' % (P, threads, n_pixels, steps)}


def cpu_worker(spec):
    import torch
    n_pixels, steps, threads = (int(x) for x in spec.split(','))
    torch.set_num_threads(threads)
    ns, dt = _cpu_steps(n_pixels, steps)
    print(json.dumps({'rate': ns * N_LIGHTS / dt}), flush=True)


# ----------------------------------------------------------------------------------------------- the data pipeline in the loop
def sampler_in_loop(device, step, steps=60, warmup=8):
    """The headline's step fed by the product's data pipeline instead of a resident batch: two BEAR-shaped views (612 x 512, 96
    lights each, ~90 %% of the pixels in the object mask) resident in HBM (handoff.DeviceViews: images as uint8, masks, stage-1 points /
    normals / visibility maps, the vis_plus tables), every step a fresh draw of light_bs = 96 lights and %d in-mask pixels in the reference's np.random
    order (stage2/datasets/dataset.py:149-151,182-185), ONE gather launch, prefetched by a worker thread.  -> sustained ms/step.""" % N_PIXELS
    import numpy as np
    import torch
    from psnerf_amd import handoff
    h, w, L = 512, 612, N_LIGHTS
    g = torch.Generator().manual_seed(7)
    views, images, omasks, lights, poses = [], [], [], [], []
    from psnerf_amd.synthetic import stage2_inputs
    for v in range(2):
        inp, _ = stage2_inputs(h * w, L, N_VIS, seed=300 + v, h=h, w=w)
        views.append({'points': inp['points'], 'normal': inp['normal'], 'surface_mask': inp['surface_mask'],
                      'visibility': inp['visibility'], 'img_res': [h, w],
                      # train.vis_plus (bear.conf:29): 256 extra supervision directions per view with their stage-1 visibility maps
                      'vis_plus': (torch.rand(256, h * w, generator=g) < 0.7).float(),
                      'vis_plus_light': torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1)})
        images.append(torch.randint(0, 256, (L, h * w, 3), generator=g, dtype=torch.uint8))   # decoded 8-bit PNGs
        omasks.append(inp['object_mask'][0])
        lights.append(inp['light_direction'])
        poses.append(inp['pose'][0])
    from psnerf_amd.stage2.trainer import VisPlus
    vp = VisPlus(views, lights, N_VIS, device)
    store = handoff.DeviceViews(views, images, omasks, lights, poses, inp['intrinsics'][0], L, device, n_pixels=N_PIXELS, vis_plus=vp)
    del images
    np.random.seed(0)
    order = [i % 2 for i in range(steps + warmup)]
    t0 = None
    feed = store.loader(order, depth=3)
    for it, (vidx, mi, gt, l_slt) in enumerate(feed):
        if it == warmup:
            settle_gc()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        # (l_slt = rows of a 2-view table; the step's light tables hold N_LIGHTS_TOTAL rows)
        terms, _ = step.step(mi, gt, l_slt, train_order=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ns = int(mi['surface_idx'].numel())
    hs = store.host_seconds
    return {'ms_per_step': round(dt * 1e3, 3), 'steps': steps, 'warmup': warmup, 'value': round(ns * L / dt, 1), 'unit': 'ray-samples/s',
            'surface_pixels_last_batch': ns, 'resident_view_bytes': store.resident_bytes(), 'image_store': 'uint8',
            'host_cpu_ms_per_item': {'draw': round(1e3 * hs['draw'] / hs['items'], 3), 'assemble': round(1e3 * hs['assemble'] / hs['items'], 3)},
            'training_thread_wait_ms_per_item': round(1e3 * feed.consumer_wait / (steps + warmup), 3),
            'batch': 'handoff.DeviceViews.loader: a different batch every step -- fresh light, pixel and vis_plus draws (V = %d of 256 + 96 '
                     'supervision directions, stage2/trainer.py:384-392), assembled on the device' % N_VIS,
            'note': 'compare VALUE with the headline (resident batch): in-mask sampling gives more surface pixels per batch than the headline batch, '
                    'so ms_per_step differ by the row count; host_cpu_ms counts thread CPU time incl. the spin of event waits (back-pressure: the '
                    'worker stays a bounded number of items ahead of the GPU); tools/run_e2e.py --full has the loop with HIP graphs '
                    '(profiles/r05*_e2e_full.json)'}


# ----------------------------------------------------------------------------------------------- stage 1 (configs[1])
def stage1_measure(device, steps=10, warmup=5, rays=4096):
    """BASELINE configs[1]: stage-1 BEAR train step, 4096 rays x 128 samples (96 inner + 32 outer, it > 5000), 256 march
    steps + 8 secant, geometric-init weights.  Per-kernel numbers from HIP events on the launch stream."""
    import torch
    from psnerf_amd import hip
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.optim import FlatAdam
    from psnerf_amd.synthetic import stage1_batch, stage1_cfg
    cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32,
                                'training.n_training_points': rays})
    it, S = 6000, 128
    batch = {k: v.to(device) for k, v in stage1_batch(cfg, h=512, w=612, seed=0).items()}
    torch.manual_seed(42)
    net = NeuralNetwork(cfg)
    ren = Renderer(net, cfg, device=device)
    tr = Trainer(ren, FlatAdam(net.parameters(), lr=1e-4), cfg, device=device)
    for _ in range(warmup):
        tr.train_step(batch, it=it)
    settle_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        terms = tr.train_step(batch, it=it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # per-kernel numbers from a second, separately instrumented pass (HIP events around every C-ABI launch)
    hip.PROFILE_EVENTS = ev = []
    for _ in range(steps):
        tr.train_step(batch, it=it)
    torch.cuda.synchronize()
    hip.PROFILE_EVENTS = None
    launches = None
    try:  # device launches of one steady-state step (as launches_per_step of the headline)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            tr.train_step(batch, it=it)
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if getattr(e, 'device_type', None) is not None and 'cuda' in str(e.device_type).lower()]
        if names:
            ours = sum(1 for n in names if 'psn::' in n)
            launches = {'total': len(names), 'hip_hand_written': ours, 'other': len(names) - ours}
    except Exception as e:  # noqa: BLE001
        launches = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}

    def agg(name):
        sel = [(u, a.elapsed_time(b), f) for (k, u, a, b, f) in ev if k == name]
        return sel

    out = {'workload': 'stage1 BEAR train step (BASELINE configs[1]): %d rays x %d samples, 256 march steps + 8 secant, '
                       'full step (march, render fwd, loss, double backward, Adam)' % (rays, S),
           'value': round(rays * S / dt, 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': steps,
           'launches_per_step': launches,
           'warmup': warmup, 'loss': round(float(terms['loss'].detach()), 6), 'dtype': 'f32', 'data': 'synthetic'}
    ch = [(u, ms, f) for u, ms, f in agg('mlp_chain') if f]
    if ch:
        fl, ms = sum(f for _, _, f in ch), sum(m for _, m, _ in ch)
        out['chain_engine'] = {'bound': 'mfma', 'kernel': 'mlp_infer_kernel<true,16>', 'achieved': round(fl / ms * 1e-9, 2),
                               'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(fl / ms * 1e-9 / PEAK_F32_MFMA_TFLOPS, 4),
                               'ms_per_step': round(ms / steps, 3), 'launches_per_step': round(len(ch) / steps, 1)}
    le = [(u, ms, f) for u, ms, f in agg('march_sweep') if f]
    if le:
        # f = (device counter of evaluated 64-step blocks, flops per block): evaluated work only (blocks behind a ray's first
        # sign change are skipped); 'dense_rows_frac' = evaluated / (N x M)
        blocks = sum(int(f[0].item()) for _, _, f in le)
        fl, ms = sum(int(f[0].item()) * f[1] for _, _, f in le), sum(m for _, m, _ in le)
        rows_dense = sum(u for u, _, _ in le)
        out['occupancy_engine'] = {'bound': 'mfma', 'kernel': 'mlp_infer_kernel<false,16,3> (march sweep: points + encoding in the prologue, early exit)',
                                   'achieved': round(fl / ms * 1e-9, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': round(fl / ms * 1e-9 / PEAK_F32_MFMA_TFLOPS, 4), 'ms_per_step': round(ms / steps, 3),
                                   'launches_per_step': round(len(le) / steps, 1),
                                   'evaluated_rows_frac': round(blocks * 64 / rows_dense, 4),
                                   'note': 'achieved counts the rows that were evaluated (true MACs of the occupancy network); the dense '
                                           'N x M sweep of the reference formulation at the same duration would read achieved / evaluated_rows_frac'}
    gm = agg('gemm_tn_grouped')
    if gm:
        fl, ms = sum(u for u, _, _ in gm), sum(m for _, m, _ in gm)
        out['weight_grad_gemm'] = {'bound': 'mfma', 'kernel': 'gemm_tn256_grouped_kernel (+128x128 tiles, + split-K reduction)',
                                   'achieved': round(fl / ms * 1e-9, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': round(fl / ms * 1e-9 / PEAK_F32_MFMA_TFLOPS, 4), 'ms_per_step': round(ms / steps, 3)}
    for name in ('composite_fwd', 'composite_bwd'):
        cp = agg(name)
        if cp:
            by, ms = sum(u for u, _, _ in cp), sum(m for _, m, _ in cp)
            out[name] = {'bound': 'hbm', 'achieved': round(by / ms * 1e-6, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': round(by / ms * 1e-6 / PEAK_HBM_GBS, 4), 'us_per_launch': round(ms / len(cp) * 1e3, 1),
                         'note': '4096 rays = 10-19 MB per launch: launch/latency-bound at this size; the HBM roofline of '
                                 'the kernel is measured at 2M rays by tools/bench_composite.py (profiles/)'}
    # EXPERIMENT, never the stage-1 number above: the gradient-free ray march of the step (rendering.py:410-523 runs under no_grad)
    # on the split-bf16 occupancy engine (psn_march_sweep_x3); every differentiated launch stays exact f32
    try:
        net.inference_precision = 'bf16x6'
        for _ in range(3):
            tr.train_step(batch, it=it)
        settle_gc()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            terms6 = tr.train_step(batch, it=it)
        torch.cuda.synchronize()
        dt6 = (time.perf_counter() - t0) / steps
        # ... and additionally the 256 x 256 weight gradients of the step on the split-bf16 kernel (psn_gemm_tn_grouped_x3)
        dts = {}
        for mode in ('bf16x6', 'bf16x3'):
            with hip.wgrad_precision(mode):
                for _ in range(3):
                    tr.train_step(batch, it=it)
                settle_gc()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    terms_m = tr.train_step(batch, it=it)
                torch.cuda.synchronize()
                dts[mode] = ((time.perf_counter() - t0) / steps, round(float(terms_m['loss'].detach()), 6))
        # ... and the matrix work of the four geometry chains and the two appearance chains as three bf16 partial products
        # (ops.chain_precision('bf16x3')) beside the three-product weight gradients: every MFMA of the step on the bf16 pipe
        # except the (latency-bound) root finder
        from psnerf_amd import ops
        net.inference_precision = 'bf16x3'   # (the ray-march sweep through the exact engine's kernel on split-bf16 weight stages)
        with ops.chain_precision('bf16x3'), hip.wgrad_precision('bf16x3'):
            for _ in range(3):
                tr.train_step(batch, it=it)
            settle_gc()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                terms_c = tr.train_step(batch, it=it)
            torch.cuda.synchronize()
            dtc = (time.perf_counter() - t0) / steps
        net.inference_precision = 'bf16x6'
        net.invalidate_packs()
        out['bf16x3_chains_experiment'] = {
            'value': round(rays * S / dtc, 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dtc * 1e3, 3), 'steps': steps,
            'loss': round(float(terms_c['loss'].detach()), 6),
            'dtype': 'f32 emulated on the bf16 matrix pipe: 2 x bf16 split operands, 3 partial products, f32 accumulate (~16 significant bits)',
            'scope': "ops.chain_precision('bf16x3') + hip.wgrad_precision('bf16x3') + inference_precision 'bf16x3': value pass, gradient sweep, "
                     'their adjoints, the appearance chains (activation programs, dumps, epilogues f32), the 256 x 256 weight gradients and the '
                     'ray-march sweep; root finder, composite, losses, Adam: exact f32.  Gates: tests/test_bf16_gpu.py (parameter gradients vs '
                     'the exact step), tests/test_convergence_gpu.py (synchronised windows)'}
        dt7, loss7 = dts['bf16x6']
        out['bf16x6_wgrad_experiment'] = {'value': round(rays * S / dt7, 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dt7 * 1e3, 3), 'steps': steps,
                                          'loss': loss7,
                                          'three_products': {'value': round(rays * S / dts['bf16x3'][0], 1), 'ms_per_step': round(dts['bf16x3'][0] * 1e3, 3),
                                                             'loss': dts['bf16x3'][1],
                                                             'scope': "hip.wgrad_precision('bf16x3'): two bf16 pieces per operand, three partial "
                                                                      'products (~16 significant bits; kernel error 5e-6 against 5e-7 of the f32 kernel)'},
                                          'dtype': 'f32 emulated on the bf16 matrix pipe (3 x bf16 split operands, 6 partial products, f32 accumulate)',
                                          'scope': 'as bf16x6_experiment (ray-march sweep) + the 256 x 256-tile weight-gradient products of the geometry and '
                                                   'appearance networks (hip.wgrad_precision); the four chains of the geometry field, the appearance chains and '
                                                   'the root finder: exact f32'}
        out['bf16x6_experiment'] = {'value': round(rays * S / dt6, 1), 'unit': 'ray-samples/s', 'ms_per_step': round(dt6 * 1e3, 3), 'steps': steps,
                                    'loss': round(float(terms6['loss'].detach()), 6),
                                    'dtype': 'f32 emulated on the bf16 matrix pipe (3 x bf16 split operands, 6 partial products, f32 accumulate)',
                                    'scope': "NeuralNetwork.inference_precision = 'bf16x6': the ray-march sweep only (gradient-free); the secant root "
                                             'finder, the render forward, both backward passes and the weight gradients: exact f32'}
    except Exception as e:  # noqa: BLE001
        out['bf16x6_experiment'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    finally:
        net.inference_precision = 'fp32'
    return out


def _stage1_cfg1(rays=512):
    """BASELINE configs[0]: stage1 BUNNY synthetic, 1 view, 512 rays x 64 samples (bunny.yaml: near 2, far 6, it = 0 -> 64 interval
    samples), 256 march steps + 8 secant; geometric-init weights, a 512 x 612 synthetic view."""
    import torch
    from psnerf_amd.synthetic import stage1_batch, stage1_cfg
    cfg = stage1_cfg('bunny', **{'training.n_training_points': rays})
    batch = stage1_batch(cfg, h=512, w=612, seed=0)
    g = torch.Generator().manual_seed(5)
    pix = torch.stack([torch.randint(0, 612, (rays,), generator=g).float(), torch.randint(0, 512, (rays,), generator=g).float()], -1)[None]
    return cfg, batch, pix, g


def stage1_cpu_baseline(steps=3):
    """BASELINE.md section 3 for stage 1 = BASELINE configs[0] itself: the oracle's Trainer.train_step (port of
    stage1/model/training.py:46-60 -- march, render forward, loss, double backward, Adam) on the host cores at 512 rays x 64 samples.
    Thread count swept on a 128-ray sample over {1, 8, 16}; the best count then runs the 512-ray step (1 warm-up + ``steps``
    timed, minimum); one thread (what the stage-2 trainer of the reference pins; stage 1 does not pin) is reported beside it."""
    import torch
    from oracle import stage1 as o1

    def run(rays, n_timed):
        cfg, batch, pix, _ = _stage1_cfg1(rays)
        torch.manual_seed(42)
        net = o1.NeuralNetwork(cfg)
        tr = o1.Trainer(o1.Renderer(net, cfg), torch.optim.Adam(net.parameters(), lr=1e-4), cfg)
        tr.train_step(batch, it=0, pix=pix)
        best = None
        for _ in range(n_timed):
            t0 = time.time()
            tr.train_step(batch, it=0, pix=pix)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        return best
    nproc = os.cpu_count() or 1
    t_all = torch.get_num_threads()
    sweep = {}
    try:
        for th in sorted({t for t in (1, 16) if t <= nproc} | {min(8, nproc)}):
            torch.set_num_threads(th)
            sweep[th] = 128 * 64 / run(128, 1)
        best_th = max(sweep, key=sweep.get)
        torch.set_num_threads(best_th)
        dt = run(512, steps)
    finally:
        torch.set_num_threads(t_all)
    return {'value': 512 * 64 / dt, 'unit': 'ray-samples/s', 'cores': best_th, 'kind': 'port', 'host_cores': nproc,
            'sample': 'oracle/stage1.py Trainer.train_step at BASELINE configs[0] (bunny, 512 rays x 64 samples, 256 march steps + 8 secant), '
                      'min of %d timed steps after 1 warm-up, %.2f s/step, %d threads = the best of the sweep' % (steps, dt, best_th),
            'thread_sweep_128rays': {str(k): round(v, 1) for k, v in sorted(sweep.items())},
            'single_thread': {'value': sweep.get(1), 'cores': 1, 'sample': '128 rays x 64 samples, 1 timed step after 1 warm-up'}}


def stage1_parity(device):
    """The metric's "PSNR parity" for stage 1 at BASELINE configs[0]: ONE full train step of the oracle (CPU) and of the HIP path from
    the same weights on the same 512 rays with the same injected draws (hit / miss jitter, neighbour offsets): rendered rgb, loss
    terms, PSNR against the synthetic ground truth, parameters after the Adam step."""
    import math
    import torch
    from oracle import stage1 as o1  # checker only
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    cfg, batch, pix, g = _stage1_cfg1(512)
    torch.manual_seed(42)
    onet = o1.NeuralNetwork(cfg)
    net = NeuralNetwork(cfg)
    net.load_state_dict(onet.state_dict())
    oren, ren = o1.Renderer(onet, cfg), Renderer(net, cfg, device=device)
    otr = o1.Trainer(oren, torch.optim.Adam(onet.parameters(), lr=1e-4), cfg)
    tr = Trainer(ren, torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=device)
    with torch.no_grad():  # the hit count decides the shapes of the draws (rendering.py:139,163,204): a dry march of the oracle
        dry = oren(pix, batch['img.camera_mat'], batch['img.world_mat'], batch['img.scale_mat'], 'unisurf', add_noise=False, eval_=True, it=0)
    n_hit = int(dry['mask_pred'].sum())
    noise = {'miss': torch.rand(1, 512 - n_hit, 64, generator=g), 'hit': torch.rand(1, n_hit, 64, generator=g), 'nbr': torch.rand(n_hit, 3, generator=g)}
    cap = {}
    h1 = oren.register_forward_hook(lambda m, i, o: cap.__setitem__('o', o))
    h2 = ren.register_forward_hook(lambda m, i, o: cap.__setitem__('p', o))
    try:
        ot = otr.train_step(batch, it=0, pix=pix, noise=noise)
        pt = tr.train_step({k: v.to(device) for k, v in batch.items()}, it=0, pix=pix, noise={k: v.to(device) for k, v in noise.items()})
    finally:
        h1.remove(); h2.remove()
    a, b = cap['p']['rgb'].detach().cpu().double().reshape(-1, 3), cap['o']['rgb'].detach().double().reshape(-1, 3)
    same_mask = bool(torch.equal(cap['p']['mask_pred'].cpu(), cap['o']['mask_pred']))
    from oracle.stage1 import gather_pixels
    gt = gather_pixels(batch['img'], pix).double().reshape(-1, 3)
    psnr = lambda x: -10.0 * math.log10(float(((x - gt) ** 2).mean()))
    ratio = (a - b).abs() / (1e-4 * b.abs() + 1e-6)
    lo, lh = float(ot['loss'].detach()), float(pt['loss'].detach())
    pd = max(float((p.detach().cpu() - q.detach()).abs().max()) for p, q in zip(net.parameters(), onet.parameters()))
    return {'sample': 'BASELINE configs[0]: bunny, 512 rays (%d hit) x 64 samples, 256 march steps + 8 secant, it = 0, one full step, same weights / '
                      'pixels / jitter draws' % n_hit,
            'hit_masks_equal': same_mask, 'rgb_max_rel_err': float((a - b).abs().max() / b.abs().max()),
            'rgb_worst_over_bound': round(float(ratio.max()), 3), 'rgb_frac_beyond_bound': float((ratio > 1).double().mean()),
            'bound': '1e-4 |ref| + 1e-6 elementwise', 'loss_hip': lh, 'loss_oracle': lo, 'loss_rel_err': abs(lh - lo) / abs(lo),
            'loss_terms_rel_err': {k: (abs(float(pt[k].detach()) - float(ot[k].detach())) / max(abs(float(ot[k].detach())), 1e-12)) for k in ot if ot[k] is not None},
            'psnr_hip_db': round(psnr(a), 5), 'psnr_oracle_db': round(psnr(b), 5), 'psnr_diff_db': round(psnr(a) - psnr(b), 6),
            'max_param_diff_after_step': pd,
            'fixture': 'tests/golden/stage1_unisurf_cfg1.npz (the reference\'s own Renderer.unisurf + Loss + backward at 512 x 64) is checked by '
                       'tests/test_stage1_gpu.py::test_unisurf_golden[cfg1]; 200-step horizon: tests/test_convergence_gpu.py'}


# ----------------------------------------------------------------------------------------------- cfg 4 (strong scaling of cfg 3)
def settle_gc():
    """Full collection + gc.freeze() in front of a timed region: the interpreter's cyclic collector stays ON, but the ~270k objects
    that exist by now (modules, torch, the model) move to the permanent generation, so that a generation-2 pass that happens to fall
    into the region walks the step's own garbage only -- one such pass over everything was measured at 70 ms
    (tools/dbg/gc_probe.py), three steps' worth of a 0.5-s region."""
    import gc
    gc.collect()
    gc.freeze()


def time_steps(fn, steps, warmup, world, device):
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    settle_gc()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, t1 - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0]) / steps * 1e3, float(t[1]) / steps * 1e3


def cfg4_case(device, dp, px_global, rank, world, steps, modes=('eager', 'graph', 'graph_fresh_batches')):
    """One stage-2 step of cfg 3's batch of ``px_global`` pixels sharded over the ranks of ``dp``: ms/step (max over ranks)
    and host issue time, eagerly and replayed from HIP graphs (fresh TrainStep each: same weights, same batch).
    'graph_fresh_batches': what a training loop sees -- a DIFFERENT batch every step (four batches with different surface masks
    and counts, cycled; the reference's dictionary without an index list), copied into the graph's input buffers, the surface
    list built on the device and padded to the pixel count (GraphedTrainStep(pad_to_pixels=True): one graph, no host sync)."""
    import torch
    import torch.distributed as dist
    from psnerf_amd.synthetic import stage2_inputs
    from psnerf_amd.stage2.graph import GraphedTrainStep
    l_slt = torch.arange(N_LIGHTS, device=device) + 96 * 3
    inp, gt = stage2_inputs(px_global, N_LIGHTS, N_VIS, seed=100, device=device, with_surface_idx=True)
    if world > 1:
        inp, gt = dp.shard_stage2(inp, gt)
    ns = torch.tensor([int(inp['surface_mask'].sum())], device=device, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(ns, op=dist.ReduceOp.SUM)
    out = {'pixels_per_gpu': inp['uv'].shape[1], 'surface_pixels_total': int(ns.item())}

    def agree(ok):  # every rank replays, or none does (a rank whose capture failed would leave the others in a collective)
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    for mode in modes:
        step = make_step(device, dp=dp)
        ns_mode = out['surface_pixels_total']
        if mode == 'eager':
            fn = lambda: step.step(inp, gt, l_slt, train_order=False)
        elif mode == 'graph_fresh_batches':
            batches = []
            for k in range(4):
                bi, bg = stage2_inputs(px_global, N_LIGHTS, N_VIS, seed=101 + k, device=device, surface_frac=0.86 + 0.02 * k)
                if world > 1:
                    bi, bg = dp.shard_stage2(bi, bg)
                batches.append((bi, bg))
            nsf = torch.tensor([sum(int(b[0]['surface_mask'].sum()) for b in batches)], device=device, dtype=torch.int64)
            if world > 1:
                dist.all_reduce(nsf, op=dist.ReduceOp.SUM)
            ns_mode = int(nsf.item()) / 4.0  # (mean surface count of the cycled batches, whole job)
            run = GraphedTrainStep(step, pad_to_pixels=True, agree=agree)
            it = [0]

            def fn():
                bi, bg = batches[it[0] % 4]
                it[0] += 1
                return run.step(bi, bg, l_slt, train_order=False)
        else:
            run = GraphedTrainStep(step, adopt_inputs=True, agree=agree)
            fn = lambda: run.step(inp, gt, l_slt, train_order=False)
        try:
            ms, host_ms = time_steps(fn, steps, 5, world, device)
        except RuntimeError as e:  # (raised on EVERY rank at the same point: see agree)
            out[mode] = {'error': str(e)[:300]}
            continue
        torch.cuda.synchronize()
        out[mode] = {'ms_per_step': round(ms, 3), 'host_issue_ms': round(host_ms, 3),
                     'value': round(ns_mode * N_LIGHTS / (ms * 1e-3), 1)}
        if mode == 'graph_fresh_batches':
            out[mode]['mean_surface_pixels_total'] = ns_mode
        del step, fn
    return out


def strong_cfg4(device, dp, rank, world, steps=40, graph_at_n=False):
    """BASELINE cfg 4 = cfg 3 (32768 px) sharded by pixels over the ranks (stage2/trainer.py:355-410 is the step
    being sharded).  At N = 1 the rank shards of N = 2, 4, 8 are timed on this GPU as well, with the data-parallel code path
    ON (a process group of one rank on RCCL): count all-reduce, flat-bucket gather + all-reduce, Adam on the bucket.
    At N > 1 only the EAGER step of the shard is timed unless ``graph_at_n`` (`bench.py --cfg4-graph`): capturing HIP graphs next to
    RCCL's threads on more than one GPU has never run anywhere (no multi-GPU box was available to the builder), and a hang there --
    although behind the watchdog -- would turn the launcher's exit code into 3; the replayed numbers of the shards are the N = 1 line's."""
    import torch
    import torch.distributed as dist
    from psnerf_amd import dist as pdist
    res = {'scaling': 'strong', 'global_pixels': N_PIXELS, 'unit': 'ray-samples/s', 'steps': steps, 'warmup': 5,
           'definition': 'BASELINE cfg 4: the 32768-px batch of cfg 3 split by pixels over N ranks, full light set per rank'}
    made_pg = False
    try:
        if world == 1 and not (dist.is_available() and dist.is_initialized()):
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', LOCAL_RANK=str(device.index or 0), WORLD_SIZE='1')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group(backend='nccl', rank=0, world_size=1)
            made_pg = True
        dpf = pdist.DataParallel(device, force=True)
        res['data_parallel_path'] = bool(dpf.enabled)
        modes = ('eager', 'graph', 'graph_fresh_batches') if (world == 1 or graph_at_n) else ('eager',)
        res['modes'] = list(modes)
        case = cfg4_case(device, dpf, N_PIXELS, rank, world, steps, modes=modes)
        res.update(case)
        best = min((m for m in ('eager', 'graph') if 'ms_per_step' in case.get(m, {})), key=lambda m: case[m]['ms_per_step'])
        res['value'], res['ms_per_step'], res['mode'] = case[best]['value'], case[best]['ms_per_step'], best
        if world == 1:
            proj = {}
            for n in (2, 4, 8):
                c = cfg4_case(device, dpf, N_PIXELS // n, rank, world, steps)
                bm = min((m for m in ('eager', 'graph') if 'ms_per_step' in c[m]), key=lambda m: c[m]['ms_per_step'])
                proj[str(n)] = {'pixels_per_rank': N_PIXELS // n, 'eager_ms': c['eager'].get('ms_per_step'), 'graph_ms': c['graph'].get('ms_per_step'),
                                'graph_fresh_batches_ms': c['graph_fresh_batches'].get('ms_per_step'),
                                'eager_host_issue_ms': c['eager'].get('host_issue_ms'), 'graph_host_issue_ms': c['graph'].get('host_issue_ms'),
                                'projected_speedup_vs_1': round(res['ms_per_step'] / c[bm]['ms_per_step'], 3),
                                'fraction_of_linear': round(res['ms_per_step'] / c[bm]['ms_per_step'] / n, 3)}
            res['per_rank_projection'] = proj
            res['projection_note'] = ('one GPU, world of one rank: RCCL executes every collective of the step, but a 1-rank all-reduce '
                                      'moves nothing -- the 2.7 MB bucket all-reduce over xGMI comes on top at N > 1 (allreduce_ms of the N > 1 lines)')
    except Exception as e:  # noqa: BLE001  (a diagnostic object must not take the headline down)
        res['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
    finally:
        if made_pg:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
    return res


def guarded(fn, line, rank, timeout=240):
    """fn() under a watchdog: if it has not returned after ``timeout`` seconds (a rank stuck in a collective cannot be interrupted),
    rank 0 prints ``line`` with an error in place of the object and every rank leaves the process with exit code 3: a run that
    hung (RCCL, graph capture) must not look like a success to the launcher, even though the headline line was printed."""
    import threading
    lock, state = threading.Lock(), {'done': False}

    def fire():
        with lock:
            if state['done']:
                return
            if rank == 0 and line is not None:
                line['strong_cfg4'] = {'error': 'strong_cfg4 did not return within %d s at this world size; the headline and every other object of '
                                                'this line were measured before it started' % timeout}
                print(json.dumps(line), flush=True)
            os._exit(3)

    t = threading.Timer(timeout, fire)
    t.daemon = True
    t.start()
    try:
        return fn()
    finally:
        with lock:
            state['done'] = True
        t.cancel()


def run_cfg4_child(timeout=900):
    """``python bench.py --cfg4-child`` as a child process; -> its strong_cfg4 dictionary, or {'error': ...}."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        r = subprocess.run([sys.executable, BENCH_PY, '--cfg4-child'], env=env, capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith('CFG4_JSON '):
                return json.loads(line[len('CFG4_JSON '):])
        return {'error': 'child exited with code %d without a result' % r.returncode, 'stderr_tail': r.stderr[-400:]}
    except Exception as e:  # noqa: BLE001
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}


