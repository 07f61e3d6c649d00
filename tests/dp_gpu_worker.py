#!/usr/bin/env python3
"""Two data-parallel ranks of the HIP product path on ONE GPU (gloo carries the collectives, both ranks use cuda:0).

    python tests/dp_gpu_worker.py OUT.json

For the stage-2 ``TrainStep`` and the stage-1 ``Trainer``: every rank takes its pixel / ray slice (ragged last shard),
runs one full optimisation step through psnerf_amd.dist.DataParallel (global loss denominators + flat-bucket
all-reduce), and rank 0 compares the all-reduced gradients -- dense MLP weights AND the light tables --, the loss terms
and the updated parameters with a single-rank step of the same HIP modules on the whole batch (SURVEY 8e:
sum_r grad_r == single-GPU gradient).  Writes {"ok": bool, "checks": {name: max-normalised error}} to OUT.json.

This file is the parent of its own rank processes and never touches the GPU itself (tests/conftest.py starts it before
pytest initialises HIP; see there).
"""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORLD = 2
TOL_GRAD = 1e-5     # sum of rank gradients vs single-rank gradient, max-normalised per tensor
TOL_LOSS = 1e-6


def _err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _stage2(rank, dev, checks):
    import torch
    import psnerf_amd.stage2 as s2
    from psnerf_amd import dist as pdist
    from psnerf_amd.synthetic import stage2_inputs
    from tests.helpers import stage2_state_dict
    from oracle import stage2 as o2  # only for the seeded, non-degenerate state dict of the parity tests
    conf = s2.bear_conf()
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    N, L, V, NL = 1001, 5, 3, 24  # 1001 pixels: ranks get 501 and 500
    g = torch.Generator().manual_seed(3)
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=g), dim=-1)
    l_slt = torch.randperm(NL, generator=g)[:L].to(dev)
    inp, gt = stage2_inputs(N, L, V, seed=17)
    surf = inp['surface_mask'][0]
    nz = torch.randn(int(surf.sum()), 3, generator=g) * 0.01
    inp = {k: v.to(dev) for k, v in inp.items()}
    gt = {k: v.to(dev) for k, v in gt.items()}

    def fresh(dp):
        net = s2.PSNetwork(conf)
        net.load_state_dict(sd)
        net.to(dev)
        st = s2.TrainStep(net, conf, NL, light_init.to(dev), dev, dp=dp)
        st.cur_iter = 5001
        return net, st

    dp = pdist.DataParallel(dev)
    assert dp.enabled and dp.world == WORLD
    net, st = fresh(dp)
    mi, gts = dp.shard_stage2(inp, gt)
    lo, hi = dp.slice_bounds(N)
    rank_of_surface = torch.cumsum(surf.long(), 0) - 1
    nz_r = nz[rank_of_surface[lo:hi][surf[lo:hi]]]
    terms, _ = st.step(mi, gts, l_slt, train_order=False, noise={'xyz': nz_r.to(dev)})
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    grads['__light_dir'] = st.light_para.weight.grad.detach().clone()
    grads['__light_int'] = st.light_inten_para.weight.grad.detach().clone()
    if rank != 0:
        return
    single = pdist.DataParallel(dev)
    single.enabled, single.world, single.rank = False, 1, 0
    net1, st1 = fresh(single)
    terms1, _ = st1.step(inp, gt, l_slt, train_order=False, noise={'xyz': nz.to(dev)})
    grads1 = {k: p.grad for k, p in net1.named_parameters() if p.grad is not None}
    grads1['__light_dir'] = st1.light_para.weight.grad
    grads1['__light_int'] = st1.light_inten_para.weight.grad
    assert sorted(grads) == sorted(grads1), (sorted(grads), sorted(grads1))
    for k in grads1:
        checks['stage2 grad ' + k] = (_err(grads[k], grads1[k]), TOL_GRAD)
    # Adam's first step moves every element by ~lr * sign(g): an element whose gradient sits at the fp32 noise floor may
    # step the other way, so the maximum is bounded by 2 lr and the bulk (mean) must agree tightly
    for (k, a), (_, b) in zip(net.state_dict().items(), net1.state_dict().items()):
        d = (a - b).abs()
        checks['stage2 param after step (max) ' + k] = (float(d.max().cpu()), 2 * 5e-4 + 1e-6)
        checks['stage2 param after step (mean) ' + k] = (float(d.mean().cpu()), 2e-5)
    checks['stage2 light table after step'] = (float((st.light_para.weight - st1.light_para.weight).abs().max().cpu()), 2 * 5e-4 + 1e-6)
    checks['stage2 light intensity after step'] = (
        float((st.light_inten_para.weight - st1.light_inten_para.weight).abs().max().cpu()), 2 * 1e-3 + 1e-6)


def _stage1(rank, dev, checks):
    import torch
    from psnerf_amd import dist as pdist
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.synthetic import stage1_batch, stage1_cfg
    from tests.helpers import stage1_state_dict
    n_rays = 129  # ranks get 65 and 64
    cfg = stage1_cfg('bunny', **{'training.n_training_points': n_rays})
    sd = stage1_state_dict(cfg, seed=21)
    batch = {k: v.to(dev) for k, v in stage1_batch(cfg, h=48, w=64, seed=4).items()}
    it = 1000
    gen = torch.Generator().manual_seed(it)
    pix = torch.stack([torch.randint(0, 64, (n_rays,), generator=gen).float(),
                       torch.randint(0, 48, (n_rays,), generator=gen).float()], -1)[None]

    def fresh(dp, flat=False):
        from psnerf_amd.optim import FlatAdam
        net = NeuralNetwork(cfg)
        net.load_state_dict(sd)
        ren = Renderer(net, cfg, device=dev)
        return net, ren, Trainer(ren, (FlatAdam if flat else torch.optim.Adam)(net.parameters(), lr=1e-4), cfg, device=dev, dp=dp)

    dp = pdist.DataParallel(dev)
    net, ren, tr = fresh(dp)
    with torch.no_grad():  # hit / miss classification of the whole ray set (identical on every rank)
        mask = ren(pix.to(dev), batch['img.camera_mat'], batch['img.world_mat'], batch['img.scale_mat'], 'unisurf',
                   add_noise=False, eval_=True, it=it)['mask_pred'].cpu()
    n_hit = int(mask.sum())
    S = cfg['rendering']['num_points_in']
    full = {'miss': torch.rand(1, n_rays - n_hit, S, generator=gen), 'hit': torch.rand(1, n_hit, S, generator=gen),
            'nbr': torch.rand(n_hit, 3, generator=gen)}
    lo, hi = dp.slice_bounds(n_rays)
    hit_rank, miss_rank = torch.cumsum(mask.long(), 0) - 1, torch.cumsum((~mask).long(), 0) - 1
    hs, ms = hit_rank[lo:hi][mask[lo:hi]], miss_rank[lo:hi][~mask[lo:hi]]
    mine = {'miss': full['miss'][:, ms], 'hit': full['hit'][:, hs], 'nbr': full['nbr'][hs]}
    terms = tr.train_step(batch, it=it, pix=pix, noise={k: v.to(dev) for k, v in mine.items()})
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}

    # The path the Trainer takes by default and the bench times: the SYNC-FREE forward (device-resident counts all-reduced as
    # tensors, rgb denominator = the pre-shard ray count), jitter ON through per-ray tables made for the WHOLE ray set (the
    # trainer slices them with the pixels).  Two steps whose global sizes differ while rank 0's shard keeps its size
    # (129 -> 65 / 64, 130 -> 65 / 65): the rgb denominator used to be cached by local size with the collective skipped on a hit.
    sf = []
    net_s, ren_s, tr_s = fresh(pdist.DataParallel(dev), flat=True)  # optim.FlatAdam on the bucket views; the single-process reference below: torch.optim.Adam
    calls = []
    inner = ren_s._unisurf_sync_free
    ren_s._unisurf_sync_free = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    steps = []
    for j, n_j in enumerate((129, 130)):
        gen_j = torch.Generator().manual_seed(70 + j)
        pix_j = torch.stack([torch.randint(0, 64, (n_j,), generator=gen_j).float(),
                             torch.randint(0, 48, (n_j,), generator=gen_j).float()], -1)[None]
        nz_j = {'full': torch.rand(n_j, S, generator=gen_j), 'nbr_full': torch.rand(n_j, 3, generator=gen_j)}
        steps.append((pix_j, nz_j))
        t_j = tr_s.train_step(batch, it=it + j, pix=pix_j, noise={k: v.to(dev) for k, v in nz_j.items()})
        sf.append(({k: float(v.detach()) for k, v in t_j.items()}, {k: p.grad.detach().clone() for k, p in net_s.named_parameters()}))
    assert len(calls) == 2, 'the DP trainer did not take the sync-free forward'
    if rank != 0:
        return
    single = pdist.DataParallel(dev)
    single.enabled, single.world, single.rank = False, 1, 0
    net1, _, tr1 = fresh(single)
    tr1.train_step(batch, it=it, pix=pix, noise={k: v.to(dev) for k, v in full.items()})
    for k, p in net1.named_parameters():
        # the stage-1 gradient passes through the double backward of a softplus(beta=100) network: the two slicings of
        # the ray set sum the same per-ray terms in a different order (fp32), so 1e-4 here instead of 1e-5
        checks['stage1 grad ' + k] = (_err(grads[k], p.grad), 1e-4)
    for (k, a), (_, b) in zip(net.state_dict().items(), net1.state_dict().items()):
        checks['stage1 param after step ' + k] = (float((a - b).abs().max().cpu()), 2 * 1e-4 + 1e-6)  # |delta| <= 2 lr (Adam, sign flips at the noise floor)
    net2, _, tr2 = fresh(single)
    for j, (pix_j, nz_j) in enumerate(steps):
        tr2.train_step(batch, it=it + j, pix=pix_j, noise={k: v.to(dev) for k, v in nz_j.items()})
        for k, p in net2.named_parameters():
            # (second step: the parameters differ by the first Adam step's sign flips at the noise floor, see below)
            checks['stage1 sync-free step %d grad %s' % (j, k)] = (_err(sf[j][1][k], p.grad), 1e-4 if j == 0 else 5e-3)
    for (k, a), (_, b) in zip(net_s.state_dict().items(), net2.state_dict().items()):
        checks['stage1 sync-free param after 2 steps ' + k] = (float((a - b).abs().max().cpu()), 2 * 2 * 1e-4 + 1e-6)


def _device_views(rank, dev, checks):
    """The training loop as tools/run_e2e.py runs it under data parallelism: every rank draws the same index lists (same np.random
    seed), GATHERS ONLY ITS PIXEL SLICE from the views resident in HBM (handoff.DeviceViews(dp=...), prefetching loader) and steps;
    rank 0 then runs the same three steps single-rank on the whole batches: first-step gradients (all-reduced) == single-rank
    gradients, parameters and light tables after three steps within Adam's sign-flip bound."""
    import numpy as np
    import torch
    import psnerf_amd.stage2 as s2
    from psnerf_amd import dist as pdist
    from psnerf_amd.handoff import DeviceViews
    from psnerf_amd.stage2.trainer import VisPlus
    from psnerf_amd.synthetic import stage2_inputs
    from tests.helpers import stage2_state_dict
    from tests.test_next_gpu import _toy_views
    from oracle import stage2 as o2
    views, init, imgs, omasks, ldirs, poses, K = _toy_views()
    for v in views:   # a real camera and surface geometry
        inp, _ = stage2_inputs(40 * 52, 1, 1, seed=1, h=40, w=52)
        v['points'], v['normal'] = inp['points'], inp['normal']
    poses, K = [inp['pose'][0]] * len(views), inp['intrinsics'][0]
    n_total = sum(l.shape[0] for l in ldirs)
    conf = s2.bear_conf(**{'train.light_bs': 4, 'train.vis_train_num': 5, 'brdf.net.xyz_jitter_std': 0.0})  # (no jitter: no per-row device draws to slice)
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    order, n_px = [0, 1, 0], 601   # ranks get 301 and 300 pixels

    def loop(dp):
        net = s2.PSNetwork(conf)
        net.load_state_dict(sd)
        net.to(dev)
        st = s2.TrainStep(net, conf, n_total, torch.cat(init).to(dev), dev, dp=dp)
        st.cur_iter = 5001
        store = DeviceViews(views, imgs, omasks, ldirs, poses, K, 4, dev, n_pixels=n_px, dp=dp, vis_plus=VisPlus(views, init, 5, dev))
        np.random.seed(31)
        first, n_seen = None, 0
        for vidx, mi, gt, l_slt in store.loader(order, depth=2):
            n_seen += mi['uv'].shape[1]
            st.step(mi, gt, l_slt, train_order=False)
            if first is None:
                first = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
                first['__light_dir'] = st.light_para.weight.grad.detach().clone()
        torch.cuda.synchronize()
        return net, st, first, n_seen

    dp = pdist.DataParallel(dev)
    net, st, g_dp, n_seen = loop(dp)
    lo, hi = dp.slice_bounds(n_px)
    assert n_seen == len(order) * (hi - lo), (n_seen, lo, hi)   # a rank never assembled more than its slice
    if rank != 0:
        return
    single = pdist.DataParallel(dev)
    single.enabled, single.world, single.rank = False, 1, 0
    net1, st1, g_1, n1 = loop(single)
    assert n1 == len(order) * n_px and sorted(g_dp) == sorted(g_1)
    for k in g_1:
        checks['device views: first-step grad ' + k] = (_err(g_dp[k], g_1[k]), TOL_GRAD)
    for (k, a), (_, b) in zip(net.state_dict().items(), net1.state_dict().items()):
        d = (a - b).abs()
        checks['device views: param after 3 steps (max) ' + k] = (float(d.max().cpu()), 3 * 2 * 5e-4 + 1e-6)
        checks['device views: param after 3 steps (mean) ' + k] = (float(d.mean().cpu()), 3e-5)
    checks['device views: light table after 3 steps'] = (float((st.light_para.weight - st1.light_para.weight).abs().max().cpu()), 3 * 2 * 5e-4 + 1e-6)


def run(rank, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(WORLD))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    from psnerf_amd import dist as pdist
    pdist.init_from_env(backend='gloo', set_device=False)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    checks = {}
    err = None
    try:
        _stage2(rank, dev, checks)
        dist.barrier()
        _stage1(rank, dev, checks)
        dist.barrier()
        _device_views(rank, dev, checks)
        dist.barrier()
    except Exception as e:  # report instead of hanging the peer
        import traceback
        err = '%s\n%s' % (e, traceback.format_exc())
    if rank == 0:
        bad = {k: v for k, v in checks.items() if not v[0] <= v[1]}
        with open(out_path, 'w') as f:
            json.dump({'ok': err is None and not bad and len(checks) > 0, 'error': err, 'n_checks': len(checks),
                       'failed': bad, 'worst': sorted(((v[0] / v[1], k) for k, v in checks.items()), reverse=True)[:8]}, f, indent=1)
    if err is not None:
        os._exit(1)
    dist.destroy_process_group()


def main():
    out_path = sys.argv[1]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    import torch.multiprocessing as mp
    mp.spawn(run, args=(port, out_path), nprocs=WORLD, join=True)


if __name__ == '__main__':
    main()
