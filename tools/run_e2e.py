#!/usr/bin/env python3
"""BASELINE configs[4] as ONE run: stage-1 training -> shape_extract (with shadow-ray visibility + vis_plus) ->
on-disk hand-off -> stage-2 joint optimisation (train_fix schedule, vis_plus supervision draw, light tables) ->
environment-map relighting on the fp32 path and on the bf16 MFMA engine.

ARMADILLO-shaped synthetic scene (the datasets are not shipped): stage1/configs/armadillo.yaml == bunny.yaml up to paths
(near 2, far 6, radius 2); stage2/confs/armadillo.conf == bear.conf up to paths, brdf.light_intensity = 4.0 and the
light intensities not being trained.  Ground truth: a Lambertian sphere of radius 0.6 (what geometric_init starts
from) under L directional lights per view.

    python tools/run_e2e.py [--h 48 --w 64 --views 2 --lights 6 --s1-steps 30 --s2-steps 40 ...]
    python -m torch.distributed.run --nproc-per-node N ... tools/run_e2e.py     (pixel / ray data parallel, dist.py)

    python tools/run_e2e.py --full --json-out profiles/r05_e2e_full.json          (configs[4] at its stated size, a few minutes)

Stage 2 is fed by handoff.DeviceViews (views resident in HBM, one gather launch per batch, draws prefetched by a worker thread;
--sampler host = the reference-shaped host construction) and reports, per leg, the SUSTAINED step rate with the sampler in the loop
next to the rate of the same step on a resident batch (what bench.py times).

Checks (assert): every loss finite; stage-1 and stage-2 losses decrease; |PSNR(bf16 relight) - PSNR(fp32 relight)| <=
0.05 dB against a common ground truth.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sphere_scene(K, c2w, h, w, lights, radius=0.6, albedo=(0.7, 0.5, 0.3), intensity=4.0):
    """Analytic ground truth of one view: images [L, h*w, 3] (row-major h*w), object mask [h*w], normals [h*w,3]."""
    import torch
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    d = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[0, 0], torch.ones_like(xs)], -1).reshape(-1, 3)  # fx for both axes (common.py:220)
    d = torch.nn.functional.normalize(d @ c2w[:3, :3].T, dim=-1)
    o = c2w[:3, 3]
    b = (d * o).sum(-1)
    disc = b * b - (o.dot(o) - radius * radius)
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp(min=0))
    n = torch.nn.functional.normalize(o + t[:, None] * d, dim=-1)
    cos = (n[None] * lights[:, None, :]).sum(-1).clamp(min=0)  # [L, hw]
    img = intensity * torch.tensor(albedo)[None, None, :] * cos[..., None] / 3.14159
    img = torch.where(hit[None, :, None], img.clamp(0, 1), torch.zeros_like(img))
    return img, hit, torch.where(hit[:, None], n, torch.zeros_like(n))


def main():
    """The chain under ops.STRICT: engine selection is loud (a fallback off the fused engines raises); restored on exit so that a
    caller that runs main() in-process (tests/test_e2e_gpu.py) keeps its own setting."""
    from psnerf_amd import ops as _ops
    with _ops.strict():
        _main()


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=48)
    ap.add_argument("--w", type=int, default=48)
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--lights', type=int, default=6)
    ap.add_argument('--light-bs', type=int, default=4)
    ap.add_argument('--vis-plus', type=int, default=8, help='extra visibility supervision directions per view')
    ap.add_argument('--vis-train-num', type=int, default=4)
    ap.add_argument('--rays', type=int, default=256)
    ap.add_argument('--s1-steps', type=int, default=30)
    ap.add_argument('--s2-steps', type=int, default=40)
    ap.add_argument('--pixels', type=int, default=1024, help='stage-2 pixels per step (in-mask sampling)')
    ap.add_argument('--envmap-h', type=int, default=8)
    ap.add_argument('--out', default=None, help='hand-off directory (default: a temporary one)')
    ap.add_argument('--graph', action='store_true',
                    help='stage 2: replay the train step from HIP graphs (psnerf_amd.stage2.graph.GraphedTrainStep, pad_to_pixels: one graph '
                         'per batch geometry although the surface count of every batch differs)')
    ap.add_argument('--full', action='store_true',
                    help='BASELINE configs[4] at its stated size: 512 x 612 views, 96 lights per view, stage 1 at cfg-2 shapes (4096 rays x 128 '
                         'samples), 256 vis_plus directions, stage 2 at cfg-3 shapes (32768 px x light_bs 96, V = 8) AND at the shipped '
                         'armadillo.conf shapes (light_bs 10, every in-mask pixel, the real train_fix schedule), 16 x 32 relight; explicit '
                         'size arguments still override')
    ap.add_argument('--sampler', choices=('device', 'host'), default='device',
                    help="stage-2 batches: 'device' = handoff.DeviceViews (views resident in HBM, one gather launch per batch, prefetching "
                         "loader), 'host' = handoff.ViewSampler (the reference's host-side item construction + upload)")
    ap.add_argument('--s1-it0', type=int, default=0, help='iteration number of the first stage-1 step (> 5000: 96 + 32 samples per ray)')
    ap.add_argument('--shipped-steps', type=int, default=0,
                    help='second stage-2 leg at the shipped shapes (light_bs 10, all in-mask pixels): iterations run with the REAL train_fix '
                         'schedule from iteration 0 (>= 5000 to cross the switch); 0 = skip')
    ap.add_argument('--json-out', default=None, help='also write the JSON line to this file')
    ap.add_argument('--backend', default=None, help='torch.distributed backend under torchrun (default nccl = RCCL; gloo for dry runs on one GPU)')
    ap.add_argument('--single-device', action='store_true', help='dry run: every rank uses cuda:0')
    ap.add_argument('--precision', choices=('fp32', 'bf16x3'), default='fp32',
                    help="'bf16x3' = BASELINE configs[4]'s bf16 MFMA path as built here (EXPERIMENT): every large MFMA stream of both "
                         'training loops on split-bf16 weight stages / split-bf16 weight gradients (three partial products, ~1e-5 '
                         'relative): stage 1 training.chain_precision + wgrad_precision + the ray-march sweep, stage 2 train.vis_bf16x3 '
                         '+ train.chain_precision + train.wgrad_precision; shape_extract, root finder, losses, optimisers: exact fp32')
    ap.add_argument('--occ-precision', choices=('fp32', 'bf16x6'), default='fp32',
                    help="gradient-free occupancy queries of shape_extract (ray march sweep, shadow rays): 'bf16x6' = the opt-in "
                         'split-bf16 engine; the hand-off is then ALSO extracted with the exact engine and compared')
    args = ap.parse_args()
    if args.full:
        given = set(a.split('=')[0] for a in sys.argv[1:] if a.startswith('--'))
        for flag, dest, val in (('--h', 'h', 512), ('--w', 'w', 612), ('--views', 'views', 2), ('--lights', 'lights', 96), ('--light-bs', 'light_bs', 96),
                                ('--vis-plus', 'vis_plus', 256), ('--vis-train-num', 'vis_train_num', 8), ('--rays', 'rays', 4096),
                                ('--s1-steps', 's1_steps', 150), ('--s1-it0', 's1_it0', 5001), ('--s2-steps', 's2_steps', 400), ('--pixels', 'pixels', 32768),
                                ('--envmap-h', 'envmap_h', 16), ('--shipped-steps', 'shipped_steps', 6000)):
            if flag not in given:
                setattr(args, dest, val)
        args.graph = True

    import numpy as np
    import torch
    from psnerf_amd import dist as pdist, handoff, metrics
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.optim import FlatAdam
    from psnerf_amd.synthetic import stage1_camera, stage1_cfg, look_at_pose
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2 import relight
    from psnerf_amd.stage2.trainer import VisPlus

    rank, local, world = pdist.init_from_env(backend=args.backend, set_device=not args.single_device)
    if args.single_device:
        local = 0
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.manual_seed(42)
    np.random.seed(42)  # stage1/train.py:14-15; every rank draws the same pixels / lights
    h, w = args.h, args.w
    t_all = time.time()

    # ------------------------------------------------------------------ scene
    x3 = args.precision == 'bf16x3'
    cfg = stage1_cfg('bunny', **{'training.n_training_points': args.rays, 'training.normal_loss': False,
                                 **({'training.chain_precision': 'bf16x3', 'training.wgrad_precision': 'bf16x3'} if x3 else {}),
                                 **({'rendering.num_points_in': 96, 'rendering.num_points_out': 32} if args.full else {})})
    K, _, S = stage1_camera(cfg, h=h, w=w)
    g = torch.Generator().manual_seed(1)
    poses = [look_at_pose(4.0, az_deg=25.0 * v, el_deg=15.0) for v in range(args.views)]
    lights, images, omasks, gt_normals = [], [], [], []
    for v in range(args.views):
        toward = torch.nn.functional.normalize(poses[v][:3, 3], dim=0)
        l = torch.nn.functional.normalize(toward[None] + 0.6 * torch.randn(args.lights, 3, generator=g), dim=-1)
        img, hit, _n = sphere_scene(K[0], poses[v], h, w, l)
        # 8-bit images as a decoded PNG gives them (stage2/datasets/dataset.py:121: uint8 / 255.)
        img = torch.from_numpy(np.rint(img.numpy() * 255.0).astype(np.uint8).astype(np.float32) / 255.)
        lights.append(l)
        images.append(img)
        omasks.append(hit)
        gt_normals.append(_n)

    # ------------------------------------------------------------------ stage 1 (a16): train on the mean-light image
    net1 = NeuralNetwork(cfg)
    ren = Renderer(net1, cfg, device=dev)
    tr1 = Trainer(ren, FlatAdam(net1.parameters(), lr=1e-4), cfg, device=dev)
    batches = []
    for v in range(args.views):
        mean_img = images[v].mean(0).reshape(h, w, 3).permute(2, 0, 1)[None]
        mean_img = torch.where(omasks[v].reshape(1, 1, h, w), mean_img, torch.ones_like(mean_img))  # white background
        batches.append({'img': mean_img.to(dev), 'img.mask': omasks[v].reshape(1, h, w).float().to(dev),
                        'img.world_mat': poses[v][None].to(dev), 'img.camera_mat': K.to(dev), 'img.scale_mat': S.to(dev)})
    s1_losses = []
    if x3:
        net1.inference_precision = 'bf16x3'   # the ray-march sweep of the training steps
    t0 = time.time()
    for it in range(args.s1_steps):
        if it == 5:  # (a generation-2 pass of the cyclic collector over the whole process is ~70 ms)
            import gc
            gc.collect()
            gc.freeze()
        terms = tr1.train_step(batches[it % args.views], it=args.s1_it0 + it)
        s1_losses.append(terms['loss'].detach().clone())   # (read after the loop: no host synchronisation per step)
    torch.cuda.synchronize()
    t_s1 = time.time() - t0
    net1.inference_precision = 'fp32'   # (shape_extract below: exact)
    s1_hist = torch.stack(s1_losses)
    if world > 1:   # a rank's loss is ITS share (local sums over the global denominators): the job's loss is the sum over ranks
        torch.distributed.all_reduce(s1_hist)
    s1_losses = [float(x) for x in s1_hist.cpu()]
    assert all(np.isfinite(s1_losses)), s1_losses
    k = max(2, args.s1_steps // 5)
    assert np.mean(s1_losses[-k:]) < np.mean(s1_losses[:k]), ('stage-1 loss did not decrease', s1_losses[:k], s1_losses[-k:])

    # ------------------------------------------------------------------ shape_extract + hand-off (a14, f2)
    out_dir = args.out or tempfile.mkdtemp(prefix='psnerf_e2e_')
    t0 = time.time()
    plus_dirs = []
    for v in range(args.views):
        # shape_extract.py:118-131: extra directions on the camera-facing hemisphere (farthest-point sampling there;
        # seeded random directions here -- the choice of directions is not on the accelerated path)
        toward = torch.nn.functional.normalize(poses[v][:3, 3], dim=0)
        pd = torch.nn.functional.normalize(torch.randn(args.vis_plus, 3, generator=g), dim=-1)
        pd = torch.where(((pd * toward).sum(-1) < 0)[:, None], -pd, pd)
        plus_dirs.append(pd)
        if rank == 0:
            handoff.export_view(ren, K.to(dev), poses[v][None].to(dev), S.to(dev), h, w, out_dir, v + 1,
                                light_dir=lights[v].to(dev), vis_plus_dir=pd.to(dev))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t_extract = time.time() - t0
    occ_cmp = None
    if args.occ_precision == 'bf16x6':
        # the same extraction on the split-bf16 occupancy engine (opt-in): the stage-2 run below consumes THIS hand-off; gates:
        # identical masks, surface points / normals / visibility within the exact path's parity bounds, visibility maps > 80 dB apart
        exact = [handoff.load_view(out_dir, v + 1) for v in range(args.views)]
        x3_dir = out_dir + '_bf16x6'
        net1.inference_precision = 'bf16x6'
        t0 = time.time()
        if rank == 0:
            for v in range(args.views):
                handoff.export_view(ren, K.to(dev), poses[v][None].to(dev), S.to(dev), h, w, x3_dir, v + 1,
                                    light_dir=lights[v].to(dev), vis_plus_dir=plus_dirs[v].to(dev))
        net1.inference_precision = 'fp32'
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t_x3 = time.time() - t0
        got = [handoff.load_view(x3_dir, v + 1) for v in range(args.views)]
        worst = {'points': 0.0, 'normal': 0.0, 'visibility': 0.0, 'vis_plus': 0.0}
        vis_psnr = []
        flips = 0
        for a, b in zip(exact, got):
            # a ray whose sweep value sits within 1e-6 of the threshold may be classified the other way: counted, bounded
            both = (a['surface_mask'] & b['surface_mask']).reshape(-1)
            flips += int((a['surface_mask'] != b['surface_mask']).sum())
            for key in worst:
                x, y = a[key].float(), b[key].float()
                x, y = (x[0][both], y[0][both]) if key in ('points', 'normal') else (x[:, both], y[:, both])
                if x.numel():
                    worst[key] = max(worst[key], float((x - y).abs().max()))
            vis_psnr.append(metrics.PSNR(b['visibility'].float()[:, both].numpy(), a['visibility'].float()[:, both].numpy()))
        assert flips <= max(1, int(1e-3 * args.views * h * w)), 'bf16x6 occupancy engine: %d pixels classified differently' % flips
        assert worst['points'] <= 2e-4 and worst['normal'] <= 2e-3 and worst['visibility'] <= 1e-3 and worst['vis_plus'] <= 1e-3, worst
        assert min(vis_psnr) >= 80.0, vis_psnr
        occ_cmp = {'max_abs_diff_vs_exact': {k2: float('%.3g' % v2) for k2, v2 in worst.items()}, 'mask_pixels_differing': flips,
                   'visibility_psnr_vs_exact_db': round(min(vis_psnr), 1), 'extract_seconds': round(t_x3, 2),
                   'exact_extract_seconds': round(t_extract, 2)}
        out_dir = x3_dir
    views = [handoff.load_view(out_dir, v + 1) for v in range(args.views)]
    n_surf = [int(vw['surface_mask'].sum()) for vw in views]
    assert min(n_surf) > 0, 'stage-1 surface is empty in some view'

    # ------------------------------------------------------------------ stage 2 (a19-a24)
    # "predicted" initial lights = ground truth + noise (SDPS-Net estimates are not on the path)
    light_pred = [torch.nn.functional.normalize(l + 0.05 * torch.randn(l.shape, generator=g), dim=-1) for l in lights]
    n_total = sum(l.shape[0] for l in lights)

    def light_error_deg(step):  # trainer.py:412-415
        est = torch.nn.functional.normalize(step.light_para.weight.detach(), dim=-1).cpu()
        return float(torch.rad2deg(torch.acos((est * torch.cat(lights)).sum(-1).clamp(-1, 1))).mean())

    def stage2_leg(name, light_bs, n_pixels, n_steps, switch, real_schedule, seed):
        """One stage-2 optimisation from fresh weights on the hand-off: ``n_steps`` iterations; ``real_schedule``: train_fix runs from
        iteration 0 (the switch falls at 5000 as in trainer.py:485-513), else the iteration counter jumps to 5000 at step ``switch``.
        Returns (report, net, step, runner)."""
        conf = s2.bear_conf(**{'brdf.light_intensity': 4.0, 'train.light_inten_train': False, 'train.light_bs': light_bs,
                               'train.vis_train_num': args.vis_train_num,
                               **({'train.vis_bf16x3': True, 'train.chain_precision': 'bf16x3', 'train.wgrad_precision': 'bf16x3'} if x3 else {})})
        torch.manual_seed(seed)
        np.random.seed(seed)
        net2 = s2.PSNetwork(conf).to(dev)
        vp = VisPlus(views, light_pred, args.vis_train_num, dev)
        on_device = args.sampler == 'device'
        step = s2.TrainStep(net2, conf, n_total, torch.cat(light_pred).to(dev), dev, vis_plus=None if on_device else vp)
        step.cur_iter = 0
        runner = step
        if args.graph:
            from psnerf_amd.stage2.graph import GraphedTrainStep
            agree = None
            if step.dp.enabled:   # every rank replays or every rank raises (a capture that fails on one rank must not strand the others)
                def agree(ok):
                    t = torch.tensor([1 if ok else 0], device=dev)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
                    return bool(t.item())
            runner = GraphedTrainStep(step, warmup=2, pad_to_pixels=True, agree=agree)
        order = [it % args.views for it in range(n_steps)]
        if on_device:
            store = handoff.DeviceViews(views, images, omasks, lights, poses, K[0], light_bs, dev, n_pixels=n_pixels, dp=step.dp, vis_plus=vp)
            feed = store.loader(order, depth=3)
        else:
            store = handoff.ViewSampler(views, images, omasks, lights, poses, K[0], light_bs=light_bs, n_pixels=n_pixels)

            def host_feed():
                for v in order:
                    vidx, mi, gt, l_slt = store.batch(v, device=dev)
                    if step.dp.enabled:
                        mi, gt = step.dp.shard_stage2(mi, gt)
                    yield vidx, mi, gt, l_slt
            feed = host_feed()
        err0 = light_error_deg(step)
        hist, t_marks = [], {}
        t_begin = time.time()
        w_lo = (switch + 10) if not real_schedule else max(n_steps - 600, 5010 if n_steps > 5600 else 10)   # timed window: steady state of the last phase
        last = None
        for it, (vidx, mi, gt, l_slt) in enumerate(feed):
            if it == 5:
                import gc
                gc.collect()
                gc.freeze()
            if not real_schedule and it == switch:
                step.cur_iter = 5000  # jump to the train_fix switch (trainer.py:485-513) instead of running 5000 iterations
            if it == w_lo:
                torch.cuda.synchronize()
                t_marks['lo'] = time.time()
            terms, _ = runner.step(mi, gt, l_slt, train_order=True, vidx=None if on_device else vidx)
            hist.append(terms['total'].detach().clone())   # read after the loop: the step has no host synchronisation of its own
            last = (mi, gt, l_slt)
        torch.cuda.synchronize()
        t_end = time.time()
        sustained = (n_steps - w_lo) / (t_end - t_marks['lo'])
        # the same step on a RESIDENT batch (what bench.py times): the last batch again and again
        mi, gt, l_slt = last
        mi = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in mi.items()}
        gt = {k: v.clone() for k, v in gt.items()}
        l_slt = l_slt.clone()
        k_res = min(60, max(20, n_steps // 10))
        for _ in range(5):
            runner.step(mi, gt, l_slt, train_order=True)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(k_res):
            runner.step(mi, gt, l_slt, train_order=True)
        torch.cuda.synchronize()
        resident = k_res / (time.time() - t0)
        hist_t = torch.stack(hist)
        if step.dp.enabled:   # (a rank's loss is its share of the global loss)
            torch.distributed.all_reduce(hist_t)
        losses = [float(x) for x in hist_t.cpu()]
        assert all(np.isfinite(losses)), losses[:10]
        sw = switch if not real_schedule else min(5000, n_steps)
        k = max(2, min(sw, n_steps - sw if n_steps > sw else sw) // 4)
        ph1, ph2 = losses[:sw], losses[sw:]
        assert np.mean(ph1[-k:]) < np.mean(ph1[:k]), (name, 'phase-1 loss did not decrease', ph1[:k], ph1[-k:])
        if ph2:
            assert np.mean(ph2[-k:]) < np.mean(ph2[:k]), (name, 'phase-2 loss did not decrease', ph2[:k], ph2[-k:])
        n_batch = int(last[0]['uv'].shape[1])
        rep = {'sampler': args.sampler, 'light_bs': light_bs, 'pixels_per_step_per_rank': n_batch, 'steps': n_steps,
               'schedule': 'train_fix from iteration 0 (switch at 5000)' if real_schedule else 'iteration counter set to 5000 at step %d' % switch,
               'loss_phase1': [ph1[0], ph1[-1]], 'loss_phase2': [ph2[0], ph2[-1]] if ph2 else None, 'seconds': round(t_end - t_begin, 2),
               'sustained_steps_per_s': round(sustained, 2), 'sustained_window_steps': n_steps - w_lo,
               'resident_batch_steps_per_s': round(resident, 2), 'sustained_over_resident': round(sustained / resident, 4),
               'ray_samples_per_s_sustained': round(sustained * n_batch * light_bs * world, 1),
               'light_direction_error_deg': [round(err0, 3), round(light_error_deg(step), 3)]}
        if on_device:
            hs = store.host_seconds
            rep['device_sampler'] = {'resident_view_bytes': store.resident_bytes(), 'image_store': str(store.tables[0]['images'].dtype),
                                     'host_ms_per_item': {'draw (np.random, worker thread)': round(1e3 * hs['draw'] / max(hs['items'], 1), 3),
                                                          'assemble (index upload + 1 gather launch, worker thread)': round(1e3 * hs['assemble'] / max(hs['items'], 1), 3)},
                                     'training_thread_wait_ms_per_item': round(1e3 * feed.consumer_wait / max(n_steps, 1), 3)}
        if args.graph:
            rep['graph'] = {'captures': runner.n_captures, 'replays': runner.n_replays, 'eager_steps': runner.n_eager}
        return rep, net2, step, runner

    switch = args.s2_steps // 2
    rep2, net2, step, runner = stage2_leg('cfg3', args.light_bs, args.pixels, args.s2_steps, switch, False, 43)
    t_s2 = rep2['seconds']
    ph1, ph2 = rep2['loss_phase1'], rep2['loss_phase2']
    host_probe = None
    if args.full and rank == 0:
        # what the reference-shaped host sampler costs per batch at this size (two items; never in a timed region above)
        hs_ = handoff.ViewSampler(views, images, omasks, lights, poses, K[0], light_bs=args.light_bs, n_pixels=args.pixels)
        t0 = time.time()
        for v in (0, 1):
            hs_.batch(v, device=dev)
        torch.cuda.synchronize()
        host_probe = {'host_sampler_seconds_per_batch': round((time.time() - t0) / 2, 3), 'torch_threads': torch.get_num_threads()}
    shipped = None
    if args.shipped_steps > 0:
        # the shapes the shipped configuration trains with (armadillo.conf: light_bs 10, train_all_pixels + sample_in_mask = every in-mask
        # pixel of the view in a fresh order), the real schedule; this leg's model is the one evaluated and relit below
        rep_s, net2, step, runner = stage2_leg('shipped', 10, h * w, args.shipped_steps, 5000, True, 44)
        # quality against the analytic ground truth: view 0 under its own (optimised) lights through evaluate()'s loop (relight.render_view)
        test_store = handoff.DeviceViews(views, images, omasks, lights, poses, K[0], args.lights, dev, n_pixels=None, split='test')
        _, mi_t, gt_t, _ = test_store.batch(0)
        base_t = {k2: mi_t[k2] for k2 in ('uv', 'intrinsics', 'pose', 'object_mask', 'normal', 'points', 'surface_mask')}
        net2.eval()
        ld_t, li_t = relight.eval_lights(None, torch.arange(args.lights, device=dev), step.light_para, None, light_offset=0)
        maps = relight.render_view(net2, base_t, ld_t, li_t, light_batch=32)
        m = (mi_t['surface_mask'][0] & mi_t['object_mask'][0])
        mse = float(((maps['rgb'][:, m] - gt_t['rgb'][:, m]) ** 2).mean())
        n_gt = gt_normals[0].to(dev)
        cosn = (torch.nn.functional.normalize(maps['normal'][m], dim=-1) * n_gt[m]).sum(-1).clamp(-1, 1)
        rep_s['view0_quality'] = {'psnr_db_vs_ground_truth_images_96_lights': round(-10.0 * np.log10(mse), 3),
                                  'normal_mae_deg_vs_analytic_sphere': round(float(torch.rad2deg(torch.acos(cosn)).mean()), 3),
                                  'surface_pixels': int(m.sum())}
        net2.train()
        shipped = rep_s
        del test_store, maps, mi_t, gt_t

    # ------------------------------------------------------------------ envmap relight, fp32 vs bf16 engine (f1, g1)
    net2.eval()
    lh = args.envmap_h
    env = np.random.RandomState(0).rand(lh, 2 * lh, 3).astype(np.float32) * (4.0 / (lh * 2 * lh))
    if args.sampler == 'device':   # a test-split item: every pixel of the view, no light / pixel draw (dataset.py:149-151,182)
        _, mi, _, _ = handoff.DeviceViews(views, images, omasks, lights, poses, K[0], args.light_bs, dev, n_pixels=None, split='test').batch(0)
    else:
        _, mi, _, _ = handoff.ViewSampler(views, images, omasks, lights, poses, K[0], light_bs=args.light_bs, split='test').batch(0, device=dev)
    base = {k2: mi[k2] for k2 in ('uv', 'intrinsics', 'pose', 'object_mask', 'normal', 'points', 'surface_mask')}
    relight.render_envmap(net2, base, env, light_h=min(lh, 2), light_batch=64)  # (workspaces, packs)
    torch.cuda.synchronize()
    t0 = time.time()
    rgb32 = relight.render_envmap(net2, base, env, light_h=lh, light_batch=64)
    torch.cuda.synchronize()
    t_r32 = time.time() - t0
    rgb16 = relight.render_envmap(net2, base, env, light_h=lh, light_batch=64, precision='bf16')
    torch.cuda.synchronize()
    t_relight = time.time() - t0
    noise = 0.03 * torch.randn(rgb32.shape, generator=torch.Generator().manual_seed(0)).to(dev)
    gt_img = (rgb32 + noise).clamp(0, 1).cpu().numpy()
    p32, p16 = metrics.PSNR(rgb32.cpu().numpy(), gt_img), metrics.PSNR(rgb16.cpu().numpy(), gt_img)
    assert abs(p32 - p16) <= 0.05, (p32, p16)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        line = json.dumps({
            'e2e': 'ok', 'n_gpus': world, 'image': [h, w], 'views': args.views, 'lights_per_view': args.lights,
            'surface_pixels': n_surf, 'stage1': {'steps': args.s1_steps, 'loss_first': s1_losses[0], 'loss_last': s1_losses[-1],
                                                 'seconds': round(t_s1, 2)},
            'precision': args.precision, 'shape_extract_seconds': round(t_extract, 2), 'handoff_dir': out_dir, 'occ_bf16x6': occ_cmp,
            'stage2_graph': rep2.get('graph'),
            'stage2': dict(rep2, switch_at=switch), 'stage2_shipped_shapes': shipped, 'host_sampler_probe': host_probe,
            'relight': {'envmap': [lh, 2 * lh], 'psnr_fp32': round(p32, 4), 'psnr_bf16': round(p16, 4),
                        'psnr_between': round(metrics.PSNR(rgb16.cpu().numpy(), rgb32.cpu().numpy()), 2),
                        'seconds': round(t_relight, 2), 'seconds_fp32': round(t_r32, 3), 'seconds_bf16': round(t_relight - t_r32, 3)},
            'total_seconds': round(time.time() - t_all, 2)})
        print(line)
        if args.json_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
            with open(args.json_out, 'w') as f:
                f.write(line + '\n')


if __name__ == '__main__':
    main()
