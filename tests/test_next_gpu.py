"""envmap relight eval, stage1->stage2 hand-off, checkpoint compatibility (GPU)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import ATOL_DEPTH, ATOL_UNIT, assert_close, assert_outputs_close, stage1_state_dict, stage2_state_dict, stage1_cfg
from psnerf_amd.synthetic import stage2_inputs, stage1_camera

pytestmark = pytest.mark.gpu


def test_envmap_relight_vs_oracle(cuda):
    """16x32-style lat-long relighting (here 4x8 lights) with RGB light intensities, summed over lights."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2 import relight
    from oracle import stage2 as o2
    sd = stage2_state_dict(o2.bear_conf(), seed=12)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda).eval()
    N, lh = 700, 4
    inp, _ = stage2_inputs(N, 1, 1, seed=3)
    base = {k: inp[k] for k in ('uv', 'intrinsics', 'pose', 'object_mask', 'normal', 'points', 'surface_mask')}
    env = np.random.RandomState(0).rand(lh, 2 * lh, 3).astype(np.float32) * 0.2
    lxyz, _ = relight.gen_light_xyz(lh, 2 * lh, envmap_radius=1)
    with torch.no_grad():
        ref = torch.zeros(N, 3)
        mi = dict(base)
        mi['light_direction'] = torch.nn.functional.normalize(torch.from_numpy(lxyz.reshape(-1, 3)).float(), dim=-1)
        mi['light_intensity'] = torch.from_numpy(env.reshape(-1, 3))
        ref = onet(mi)['sg_rgb_values'].sum(0).clamp(0, 1)
    base_d = {k: v.to(cuda) for k, v in base.items()}
    out = relight.render_envmap(net, base_d, env, light_h=lh, light_batch=5)
    assert_close(out.cpu(), ref, 1e-4, 'relit rgb', atol=ATOL_UNIT)
    out2, vis = relight.render_envmap(net, base_d, env, light_h=lh, light_batch=32, pixel_chunk=256, visibility=True)
    assert_close(out2.cpu(), ref, 1e-4, 'relit rgb (chunked)', atol=ATOL_UNIT)
    assert vis.shape == (N, 3)
    # the loop's short cut (PSNetwork._eval_outputs: lights summed on the surface rows, no other dense output written) against
    # the plain formulation (every dense output of the model, out['sg_rgb_values'].sum(0)): same image, same visibility
    class Plain(object):  # hides _eval_outputs from render_envmap, forwards everything else
        def __init__(self, m):
            object.__setattr__(self, 'm', m)
        def __getattr__(self, k):
            if k == '_eval_outputs':
                raise AttributeError(k)
            return getattr(object.__getattribute__(self, 'm'), k)
        def __setattr__(self, k, v):
            setattr(object.__getattribute__(self, 'm'), k, v)
        def __call__(self, *a, **kw):
            return object.__getattribute__(self, 'm')(*a, **kw)
    out3, vis3 = relight.render_envmap(Plain(net), base_d, env, light_h=lh, light_batch=32, pixel_chunk=256, visibility=True)
    assert net._eval_outputs is None and net._eval_cache is None
    assert_close(out2.cpu(), out3.cpu(), 1e-6, 'short cut vs plain loop', atol=1e-7)
    assert torch.equal(vis, vis3)


def test_handoff_roundtrip_and_checkpoints(cuda, tmp_path):
    """stage-1 shape export in the reference's npy layout, read back as stage-2 inputs; stage-1 / stage-2
    checkpoints in the reference's file formats."""
    from psnerf_amd import handoff, checkpoints
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    import psnerf_amd.stage2 as s2
    from oracle import stage1 as o1
    cfg = stage1_cfg('bunny')
    net = NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=11))
    ren = Renderer(net, cfg, device=cuda)
    h = w = 24
    K, c2w, S = stage1_camera(cfg, h=h, w=w)
    ldir = torch.nn.functional.normalize(torch.randn(3, 3, generator=torch.Generator().manual_seed(0)), dim=-1)
    pdir = torch.nn.functional.normalize(torch.randn(2, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
    out_dir = str(tmp_path / 'shape')
    mask = handoff.export_view(ren, K.to(cuda), c2w.to(cuda), S.to(cuda), h, w, out_dir, 3, light_dir=ldir.to(cuda),
                               vis_plus_dir=pdir.to(cuda), chunk=200)
    assert mask.shape == (h, w) and mask.any()
    view = handoff.load_view(out_dir, 3)
    assert view['points'].shape == (1, h * w, 3) and view['surface_mask'].shape == (1, h * w)
    assert view['visibility'].shape == (3, h * w) and view['vis_plus'].shape == (2, h * w)
    assert_close(view['vis_plus_light'], pdir, 1e-6, 'vis_plus dirs')
    # the oracle's shape_extract on the same pixel grid, mapped through the reference's to_hw
    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(stage1_state_dict(cfg, seed=11))
    pix = handoff.arange_pixels(h, w, 'cpu').float()
    o = o1.Renderer(onet, cfg)(pix, K, c2w, S, 'shape_extract', visibility=True, light_dir=torch.cat([ldir, pdir]))
    assert_close(view['points'].reshape(h, w, 3), handoff.to_hw(o['points'], h, w), 1e-4, 'points', atol=ATOL_DEPTH)
    assert np.array_equal(view['surface_mask'].reshape(h, w).numpy(), handoff.to_hw(o['mask'], h, w)[..., 0].numpy())
    v_ref = o['visibility'][:3].numpy().reshape(3, h, w).transpose(0, 2, 1).reshape(3, -1)
    assert_close(view['visibility'], v_ref, 1e-4, 'visibility', atol=ATOL_UNIT)

    # stage-1 checkpoint file (reference format) round trip
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    io = checkpoints.CheckpointIO(str(tmp_path / 'models'), model=net, optimizer=opt)
    io.save('model.pt', epoch_it=3, it=1234, loss_val_best=0.5)
    raw = torch.load(str(tmp_path / 'models' / 'model.pt'))
    assert set(raw.keys()) == {'model', 'optimizer', 'epoch_it', 'it', 'loss_val_best'}
    onet2 = o1.NeuralNetwork(cfg)
    onet2.load_state_dict(raw['model'])  # a reference-keyed module loads our file
    net2 = NeuralNetwork(cfg)
    io2 = checkpoints.CheckpointIO(str(tmp_path / 'models'), model=net2)
    scalars = io2.load('model.pt')
    assert scalars['it'] == 1234 and scalars['epoch_it'] == 3
    with pytest.raises(FileExistsError):
        io2.load('missing.pt')
    # stage-2 directory layout round trip
    conf = s2.bear_conf()
    m = s2.PSNetwork(conf).to(cuda)
    step = s2.TrainStep(m, conf, 8, torch.nn.functional.normalize(torch.randn(8, 3), dim=-1).to(cuda), cuda)
    ck = str(tmp_path / 'ckpt')
    checkpoints.save_stage2(step, ck, epoch=7)
    for sub in ('ModelParameters', 'SGOptimizerParameters', 'SGSchedulerParameters', 'OptimizerLightParameters', 'LightParameters'):
        assert os.path.exists(os.path.join(ck, sub, '7.pth')) and os.path.exists(os.path.join(ck, sub, 'latest.pth'))
    m2 = s2.PSNetwork(conf).to(cuda)
    step2 = s2.TrainStep(m2, conf, 8, torch.zeros(8, 3).to(cuda), cuda)
    assert checkpoints.load_stage2(step2, ck, 'latest') == 7
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(step.light_para.weight, step2.light_para.weight)


def test_material_editing_vs_oracle(cuda):
    """eval.py:233-312: re-render with a replaced albedo (``albedo_new``) and with a single specular basis lobe
    switched on (``basis_new``); forward only, against the oracle."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    sd = stage2_state_dict(o2.bear_conf(), seed=21)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda).eval()
    N, L = 500, 6
    inp, _ = stage2_inputs(N, L, 1, seed=9)
    inp_d = {k: v.to(cuda) for k, v in inp.items()}
    albedo_new = np.array([0.8, 0.3, 0.1], dtype=np.float32)
    for kw in (dict(albedo_new=albedo_new), dict(basis_new=3), dict(albedo_new=albedo_new, basis_new=7)):
        with torch.no_grad():
            ref = onet(inp, **kw)
            out = net(inp_d, **kw)
        for k in ('sg_rgb_values', 'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'sg_weight'):
            assert_outputs_close(k, out[k].cpu(), ref[k], prefix='%s: ' % sorted(kw))


def test_envmap_relight_vs_reference_pieces(cuda):
    """relight.render_envmap against tests/golden/stage2_relight.npz: the light-batch / pixel-chunk loop of stage2/eval.py:173-218
    re-assembled from the reference's OWN PSNetwork, gen_light_xyz, split_input and merge_output (tools/gen_golden.py), 4 x 8
    environment lights over a 30 x 40 view, RGB light intensities, light batches of 10 vs 7 here, pixel chunks."""
    import os
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2 import relight
    from oracle import stage2 as o2
    from tests.helpers import GOLDEN, state_dict_digest
    g = np.load(os.path.join(GOLDEN, 'stage2_relight.npz'))
    sd = stage2_state_dict(o2.bear_conf(), seed=12)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda).eval()
    hr, wr = (int(v) for v in g['hw'])
    inp, _ = stage2_inputs(hr * wr, 1, 1, seed=int(g['input_seed']), h=hr, w=wr)
    base = {'object_mask': torch.ones(1, hr * wr, dtype=torch.bool), 'uv': torch.from_numpy(g['uv'])[None], 'intrinsics': inp['intrinsics'],
            'pose': inp['pose'], 'normal': torch.ones(1, hr * wr, 3), 'points': inp['points'], 'surface_mask': inp['surface_mask']}
    base = {k: v.to(cuda) for k, v in base.items()}
    rgb, vis = relight.render_envmap(net, base, g['env'], light_h=int(g['light_h']), light_batch=7, pixel_chunk=500, visibility=True)
    assert_close(rgb.cpu().reshape(hr, wr, 3), g['rgb'], 1e-4, 'relit rgb', atol=ATOL_UNIT)
    assert_close(vis.cpu().reshape(hr, wr, 3), g['visibility'], 1e-4, 'light-averaged visibility', atol=1e-5)
    rgb6 = relight.render_envmap(net, base, g['env'], light_h=int(g['light_h']), light_batch=32, precision='bf16x6')
    assert_close(rgb6.cpu().reshape(hr, wr, 3), g['rgb'], 1e-4, 'relit rgb (bf16x6 experiment)', atol=ATOL_UNIT)


def _toy_views(n_views=2, h=40, w=52, P=6, seed=0, u8=True):
    """Views in the hand-off layout (handoff.load_view) + per-light images; u8: images are 8-bit values / 255 (a decoded PNG)."""
    g = torch.Generator().manual_seed(seed)
    Ls = [12, 9, 10][:n_views]
    views, init, imgs, omasks, ldirs, poses = [], [], [], [], [], []
    for L in Ls:
        views.append({'points': torch.randn(1, h * w, 3, generator=g), 'normal': torch.randn(1, h * w, 3, generator=g),
                      'surface_mask': torch.rand(1, h * w, generator=g) > 0.3, 'visibility': torch.rand(L, h * w, generator=g),
                      'vis_plus': torch.rand(P, h * w, generator=g), 'vis_plus_light': torch.randn(P, 3, generator=g), 'img_res': [h, w]})
        init.append(torch.randn(L, 3, generator=g) * 2.0)
        if u8:
            k = torch.randint(0, 256, (L, h * w, 3), generator=g)
            imgs.append(torch.from_numpy(k.numpy().astype(np.float32) / 255.))   # dataset.py:121
        else:
            imgs.append(torch.rand(L, h * w, 3, generator=g))
        omasks.append(torch.rand(h * w, generator=g) > 0.2)
        ldirs.append(torch.randn(L, 3, generator=g))
        poses.append(torch.randn(4, 4, generator=g))
    return views, init, imgs, omasks, ldirs, poses, torch.randn(4, 4, generator=g)


def _same_batch(a, b):
    (ia, ma, ga, la), (ib, mb, gb, lb) = a, b
    assert ia == ib and sorted(ma) == sorted(mb), (sorted(ma), sorted(mb))
    for k in ma:
        x, y = ma[k], mb[k]
        assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x.cpu(), y.cpu()), k
    assert sorted(ga) == sorted(gb) == ['rgb'] and ga['rgb'].shape == gb['rgb'].shape and torch.equal(ga['rgb'].cpu(), gb['rgb'].cpu())
    assert la.dtype == lb.dtype and torch.equal(la.cpu(), lb.cpu())


@pytest.mark.parametrize('u8,n_pixels,vis_plus', [(True, 700, True), (False, 700, False), (True, None, False), (True, 5000, True)])
def test_device_views_batch_equals_the_host_sampler_bit_for_bit(cuda, u8, n_pixels, vis_plus):
    """handoff.DeviceViews (views resident in HBM, draws on the host in the reference's order, ONE gather launch: psn_view_batch)
    against handoff.ViewSampler.batch (stage2/datasets/dataset.py:137-199 + trainer.py:364-392 on the host) from the same np.random
    seed: every key, shape, dtype and value of model_input / ground_truth / l_slt over several items of alternating views, the
    vis_plus selection included; 8-bit images are stored as uint8 and decoded through the k / 255 table (bit-identical); n_pixels
    None = a test-split item (all pixels, all lights); 5000 > in-mask pixels = the min() of dataset.py:185."""
    from psnerf_amd.handoff import DeviceViews, ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    views, init, imgs, omasks, ldirs, poses, K = _toy_views(u8=u8)
    split = 'train' if n_pixels is not None else 'test'
    vp_h = VisPlus(views, init, 5, 'cpu') if vis_plus else None
    vp_d = VisPlus(views, init, 5, cuda) if vis_plus else None
    host = ViewSampler(views, imgs, omasks, ldirs, poses, K, light_bs=4, n_pixels=n_pixels, split=split)
    store = DeviceViews(views, imgs, omasks, ldirs, poses, K, light_bs=4, device=cuda, n_pixels=n_pixels, split=split, vis_plus=vp_d)
    assert (store.tables[0]['images'].dtype == torch.uint8) == u8
    order = [0, 1, 1, 0, 1]
    np.random.seed(11)
    want = []
    for v in order:
        idx, mi, gt, l_slt = host.batch(v, device=cuda)
        if vis_plus:
            mi['light_vis_train'], mi['vis_train_gt'] = (t.to(cuda) for t in vp_h.select(idx, mi['sampling_idx'][0].cpu()))
        want.append((idx, mi, gt, l_slt))
    np.random.seed(11)
    got = [store.batch(v) for v in order]
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        _same_batch(a, b)
    # the prefetching loader (worker thread, side stream, ring of fixed slots) hands out the same items in the same order
    # (a fresh store: like the reference's data set the sampler is stateful -- min(len(sampling_idx), in-mask pixels), dataset.py:185)
    store = DeviceViews(views, imgs, omasks, ldirs, poses, K, light_bs=4, device=cuda, n_pixels=n_pixels, split=split, vis_plus=vp_d)
    np.random.seed(11)
    for b, item in zip(want, store.loader(order, depth=2)):
        torch.cuda.synchronize()
        _same_batch(item, b)
    # a consumer that leaves early stops the worker thread (it would otherwise wait for a free slot for ever)
    early = store.loader(order * 20, depth=2)
    next(early)
    early.close()
    assert not early.thread.is_alive()
    # a worker that fails while the bounded queue is full and the consumer sits in a long step: the error reaches the
    # consumer -- after the items made before it -- instead of leaving it blocked on an empty queue; `with` closes the loader
    import time
    with store.loader(list(order[:3]) + [10 ** 6] + list(order), depth=1) as bad:
        next(bad)
        time.sleep(1.5)
        with pytest.raises(Exception) as ei:
            for _ in range(10):
                next(bad)
        assert not isinstance(ei.value, StopIteration) and bad.error is ei.value
    assert not bad.thread.is_alive()


@pytest.mark.parametrize('world', [2, 8])
def test_device_views_rank_slice_equals_the_sharded_host_batch(cuda, world):
    """Under data parallelism every rank draws the same lists (same np.random seed) and gathers only ITS slice_bounds share of the
    pixel list: DeviceViews(dp=rank r).batch == DataParallel.shard_stage2(ViewSampler.batch) of rank r for every rank; the shares
    tile the full batch; the surface-pixel list of a shard comes from the host copy of the mask (== nonzero of the shard's mask)."""
    from psnerf_amd.dist import DataParallel
    from psnerf_amd.handoff import DeviceViews, ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    views, init, imgs, omasks, ldirs, poses, K = _toy_views()
    n_px = 1001   # not a multiple of the world size: the last rank's share is shorter
    host = ViewSampler(views, imgs, omasks, ldirs, poses, K, light_bs=5, n_pixels=n_px)
    vp_h, vp_d = VisPlus(views, init, 4, 'cpu'), VisPlus(views, init, 4, cuda)
    np.random.seed(5)
    idx, mi, gt, l_slt = host.batch(1, device=cuda)
    mi['light_vis_train'], mi['vis_train_gt'] = (t.to(cuda) for t in vp_h.select(idx, mi['sampling_idx'][0].cpu()))
    seen = 0
    for r in range(world):
        dp = DataParallel.__new__(DataParallel)
        dp.enabled, dp.world, dp.rank = True, world, r
        store = DeviceViews(views, imgs, omasks, ldirs, poses, K, light_bs=5, device=cuda, n_pixels=n_px, dp=dp, vis_plus=vp_d)
        np.random.seed(5)
        got = store.batch(1)
        mi_r, gt_r = dp.shard_stage2(mi, gt)
        _same_batch(got, (idx, mi_r, gt_r, l_slt))
        assert torch.equal(got[1]['surface_idx'], got[1]['surface_mask'][0].nonzero(as_tuple=True)[0])
        seen += got[1]['uv'].shape[1]
    assert seen == n_px


def test_train_step_on_device_views_equals_the_step_on_the_host_sampler(cuda):
    """Three optimisation steps fed by DeviceViews.loader (vis_plus draw made by the store) against the same steps fed by
    ViewSampler.batch + TrainStep's own vis_plus draw: identical losses and parameters -- the device-resident pipeline changes where
    the batch is assembled, not what is trained."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.handoff import DeviceViews, ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    views, init, imgs, omasks, ldirs, poses, K = _toy_views()
    for v, p in zip(views, poses):   # a real camera: look-at poses and BEAR-like intrinsics from the synthetic batch
        inp, _ = stage2_inputs(40 * 52, 1, 1, seed=1, h=40, w=52)
        v['points'], v['normal'] = inp['points'], inp['normal']
    inp, _ = stage2_inputs(40 * 52, 1, 1, seed=1, h=40, w=52)
    poses, K = [inp['pose'][0]] * len(views), inp['intrinsics'][0]
    n_total = sum(l.shape[0] for l in ldirs)
    res = {}
    for mode in ('host', 'device'):
        conf = s2.bear_conf(**{'train.light_bs': 4, 'train.vis_train_num': 5})
        net = s2.PSNetwork(conf)
        net.load_state_dict(stage2_state_dict(conf, seed=3))
        net.to(cuda)
        vp = VisPlus(views, init, 5, cuda)
        step = s2.TrainStep(net, conf, n_total, torch.cat(init).to(cuda), cuda, vis_plus=vp if mode == 'host' else None)
        step.cur_iter = 5000
        step._ori = (1.0, 0.05, 0.01, 1)
        step.train_fix()
        np.random.seed(21)
        torch.manual_seed(21)
        order, losses = [0, 1, 0], []
        if mode == 'host':
            ds = ViewSampler(views, imgs, omasks, ldirs, poses, K, light_bs=4, n_pixels=600)
            feed = (ds.batch(v, device=cuda) for v in order)
        else:
            feed = DeviceViews(views, imgs, omasks, ldirs, poses, K, light_bs=4, device=cuda, n_pixels=600, vis_plus=vp).loader(order)
        for vidx, mi, gt, l_slt in feed:
            terms, _ = step.step(mi, gt, l_slt, train_order=False, vidx=vidx if mode == 'host' else None)
            losses.append(float(terms['total']))
        res[mode] = (losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, step.light_para.weight.detach().cpu().clone())
    assert res['host'][0] == res['device'][0], (res['host'][0], res['device'][0])
    for k in res['host'][1]:
        assert torch.equal(res['host'][1][k], res['device'][1][k]), k
    assert torch.equal(res['host'][2], res['device'][2])


@pytest.mark.parametrize('edit', [False, True])
def test_render_view_vs_reference_pieces(cuda, edit):
    """f1: relight.render_view on the HIP model against tests/golden/stage2_eval_view.npz -- the test-view render of
    stage2/eval.py:314-417 (rgb / rough / mask / normal / albedo / visibility maps of a view under its own optimised lights, light
    batches x 1024-pixel chunks) and the material-edit loop (:233-312) assembled from the reference's own pieces."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2 import relight
    from tests.test_oracle_golden import _eval_view_case
    g, net, mi, ld, li = _eval_view_case(s2.PSNetwork, s2.bear_conf, dev=cuda)
    kw, tag = {}, 'view_'
    if edit:
        an, bn, _ = relight.edit_material(color=str(g['color']), basis=int(g['basis']), edit_albedo=True, edit_specular=True)
        kw, tag = {'albedo_new': an, 'basis_new': bn}, 'edit_'
    hv, wv = (int(v) for v in g['hw'])
    for chunk, lb in ((1024, int(g['light_batch'])), (None, 64)):
        maps = relight.render_view(net, mi, ld, li, light_batch=lb, pixel_chunk=chunk, **kw)
        for k in ('rgb', 'rough', 'visibility'):
            assert_close(maps[k].reshape(-1, hv, wv, 3).cpu(), g[tag + k], 1e-4, k, atol=ATOL_UNIT)
        for k in ('normal', 'albedo'):
            assert_close(maps[k].reshape(hv, wv, 3).cpu(), g[tag + k], 1e-4, k, atol=ATOL_UNIT)
        assert np.array_equal(maps['mask'].reshape(hv, wv).cpu().numpy(), g[tag + 'mask'].astype(bool))


def test_validate_view_vs_oracle(cuda):
    """stage2.trainer.validate_view = the validation half of TrainRunner.plot_to_disk (stage2/trainer.py:257-326): PSNR over the
    masked pixels and the normal MAE of one view under one light, whole image and in 1024-pixel chunks, against the same formulas on
    the oracle's outputs."""
    import math
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2.trainer import validate_view
    from oracle import stage2 as o2
    conf = o2.bear_conf(**{'brdf.net.xyz_jitter_std': 0})
    sd = stage2_state_dict(conf, seed=12)
    onet = o2.PSNetwork(conf)
    onet.load_state_dict(sd)
    onet.eval()
    net = s2.PSNetwork(s2.bear_conf(**{'brdf.net.xyz_jitter_std': 0}))
    net.load_state_dict(sd)
    net.to(cuda).eval()
    N = 2300
    inp, gt = stage2_inputs(N, 1, 1, seed=9)
    inp = {k: v for k, v in inp.items() if k not in ('light_vis_train', 'vis_train_gt', 'light_intensity')}
    inp['gt_normal'] = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=torch.Generator().manual_seed(2)), dim=-1)
    with torch.no_grad():
        oo = onet(inp)
    m = (oo['network_object_mask'] & oo['object_mask'])[0]
    mse = float(((oo['sg_rgb_values'][0][m] - gt['rgb'][0][m]) ** 2).mean())
    cosn = (torch.nn.functional.normalize(oo['normal_pred'][0][m], dim=-1) * inp['gt_normal'][0][m]).sum(-1).clamp(-1, 1)
    want_psnr, want_mae = -10.0 * math.log10(mse), float(torch.rad2deg(torch.acos(cosn)).mean())
    inp_d = {k: v.to(cuda) for k, v in inp.items()}
    for chunk in (None, 1024):
        rep = validate_view(net, inp_d, gt['rgb'][0].to(cuda), pixel_chunk=chunk)
        assert abs(rep['psnr'] - want_psnr) < 1e-3 and abs(rep['normal_MAE'] - want_mae) < 2e-3, (rep['psnr'], want_psnr, rep['normal_MAE'], want_mae)
        assert rep['normal_mae'].shape == (N,) and float(rep['normal_mae'][~m.to(cuda)].abs().max()) == 0.0


def test_device_views_decodes_16_bit_images_like_the_reference(cuda):
    """16-bit PNGs (imageio gives uint16; stage2/datasets/dataset.py:121 divides by 255. all the same: values up to 257): DeviceViews
    keeps the uint16 planes resident and decodes through the 65536-entry (float32) k / 255. table -- the batch equals the host sampler's
    on the float images bit for bit."""
    from psnerf_amd.handoff import DeviceViews, ViewSampler
    views, init, imgs, omasks, ldirs, poses, K = _toy_views()
    g = torch.Generator().manual_seed(9)
    k16 = [torch.randint(0, 65536, tuple(i.shape), generator=g).to(torch.int32) for i in imgs]
    imgs_f = [torch.from_numpy(k.numpy().astype(np.float32) / 255.) for k in k16]
    imgs_u16 = [torch.from_numpy(k.numpy().astype(np.uint16)) for k in k16]
    host = ViewSampler(views, imgs_f, omasks, ldirs, poses, K, light_bs=4, n_pixels=500)
    store = DeviceViews(views, imgs_u16, omasks, ldirs, poses, K, light_bs=4, device=cuda, n_pixels=500)
    assert store.tables[0]['images'].dtype == torch.uint16 and store.tables[0]['lut'].numel() == 65536
    np.random.seed(3)
    want = [host.batch(v, device=cuda) for v in (1, 0)]
    np.random.seed(3)
    got = [store.batch(v) for v in (1, 0)]
    for a, b in zip(got, want):
        _same_batch(a, b)
    assert float(got[0][2]['rgb'].max()) > 200.0   # (values / 255 of a 16-bit image, not clipped)
