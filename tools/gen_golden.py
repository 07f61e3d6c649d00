#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported read-only from
/root/reference) on seeded inputs, and cross-check the oracle against it.

Runs only in the build container (the reference never travels to the GPU box).
Usage:  python tools/gen_golden.py stage1 | stage2 | trainer | configs | all

Weights are not stored for full-width nets: tests regenerate them from seeds
(tests/helpers.py) and compare the sha256 stored in the fixture.
"""
import json
import os
import subprocess
import sys

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import (GOLDEN, grad_digest, rel_err, stage1_state_dict, stage2_state_dict,  # noqa: E402
                           state_dict_digest, stage1_cfg)
from psnerf_amd.synthetic import stage1_camera, stage1_batch, stage2_inputs  # noqa: E402

REF = '/root/reference'
torch.set_num_threads(8)


def np_(t):
    return t.detach().cpu().numpy()


def check(name, a, b, tol=1e-6):
    e = rel_err(np_(a) if torch.is_tensor(a) else a, np_(b) if torch.is_tensor(b) else b)
    flag = 'ok ' if e <= tol else 'BAD'
    print('  [%s] oracle-vs-reference %-28s rel=%.2e' % (flag, name, e))
    assert e <= tol, name


# ---------------------------------------------------------------------------
def gen_stage1():
    sys.path.insert(0, os.path.join(REF, 'stage1'))
    import model as rmdl  # the reference package, unmodified
    from model.losses import Loss as RLoss
    from oracle import stage1 as o1

    # --- init path pinned: same seed -> same weights from reference and oracle ctor
    cfg = stage1_cfg('bunny')
    torch.manual_seed(7)
    rnet = rmdl.NeuralNetwork(cfg)
    torch.manual_seed(7)
    onet = o1.NeuralNetwork(cfg)
    rsd, osd = rnet.state_dict(), onet.state_dict()
    assert sorted(rsd.keys()) == sorted(osd.keys())
    for k in rsd:
        assert torch.equal(rsd[k], osd[k]), k
    init_digest = state_dict_digest(rsd)
    print('stage1 init path identical; digest', init_digest[:16])

    for tag, over in (('h64', {'model.hidden_dim': 64, 'model.feat_size': 64}), ('h256', {})):
        cfg = stage1_cfg('bunny', **over)
        sd = stage1_state_dict(cfg, seed=11)
        rnet = rmdl.NeuralNetwork(cfg)
        rnet.load_state_dict(sd)
        onet = o1.NeuralNetwork(cfg)
        onet.load_state_dict(sd)
        g = torch.Generator().manual_seed(3)
        Q = 96
        p = torch.rand(Q, 3, generator=g) * 1.6 - 0.8
        ray_d = torch.randn(Q, 3, generator=g)
        c_rgb = torch.randn(Q, 3, generator=g)
        c_alpha = torch.randn(Q, 1, generator=g)
        c_occ = torch.randn(Q, cfg['model']['feat_size'] + 1, generator=g)
        c_grad = torch.randn(Q, 3, generator=g)

        def run(net):
            net.zero_grad()
            occ = net.infer_occ(p.clone())
            grad = net.gradient(p.clone())[:, 0]
            rgb, alpha = net(p.clone(), ray_d, return_addocc=True)
            loss = (rgb * c_rgb).sum() + (alpha * c_alpha).sum() + (occ * c_occ).sum() * 0.01 \
                + (grad * c_grad).sum() * 0.1
            loss.backward()
            grads = {k: v.grad.clone() for k, v in net.named_parameters()}
            return occ, grad, rgb, alpha, loss, grads

        r = run(rnet)
        o = run(onet)
        for nm, a, b in zip(('occ', 'grad', 'rgb', 'alpha', 'loss'), o[:5], r[:5]):
            check('%s/%s' % (tag, nm), a, b)
        names, norms, projs = grad_digest(r[5])
        _, onorms, oprojs = grad_digest(o[5])
        check('%s/param-grad norms' % tag, onorms, norms, 1e-5)
        check('%s/param-grad projs' % tag, oprojs, projs, 2e-5)
        occ_only = rnet(p, only_occupancy=True)
        logits = rnet(p, return_logits=True)
        np.savez_compressed(
            os.path.join(GOLDEN, 'stage1_net_%s.npz' % tag),
            sd_digest=state_dict_digest(sd), init_digest=init_digest,
            p=np_(p), ray_d=np_(ray_d), c_rgb=np_(c_rgb), c_alpha=np_(c_alpha), c_occ=np_(c_occ),
            c_grad=np_(c_grad), occ=np_(r[0]), grad=np_(r[1]), rgb=np_(r[2]), alpha=np_(r[3]),
            loss=np_(r[4]), occ_only=np_(occ_only), logits=np_(logits),
            grad_names=np.array(names), grad_norms=norms, grad_projs=projs,
            **({('g_' + k): np_(v) for k, v in r[5].items()} if tag == 'h64' else {}))

    # --- composite (rendering.py:196-197,214-216): formula evaluated with torch ops
    g = torch.Generator().manual_seed(5)
    comp = {}
    for S in (64, 96, 128):
        N = 37
        alpha = torch.rand(N, S, generator=g)
        alpha[0] = 0.0
        alpha[1] = 1.0
        alpha[2, : S // 2] = 0.0
        alpha[3] = 1e-7
        alpha[4, 5] = 1.0
        rgb = torch.rand(N, S, 3, generator=g)
        alpha.requires_grad_(True)
        rgb.requires_grad_(True)
        w = alpha * torch.cumprod(torch.cat([torch.ones((N, 1)), 1. - alpha + 1e-6], -1), -1)[:, :-1]
        rgb_v = torch.sum(w.unsqueeze(-1) * rgb, dim=-2)
        acc = torch.sum(w, -1)
        rgb_w = rgb_v + (1. - acc.unsqueeze(-1))
        c1 = torch.randn(N, 3, generator=g)
        c2 = torch.randn(N, generator=g)
        ((rgb_w * c1).sum() + (acc * c2).sum()).backward()
        ow, orgb = o1.alpha_composite(alpha.detach(), rgb.detach())
        check('composite S=%d' % S, orgb, rgb_v, 1e-7)
        comp.update({'alpha%d' % S: np_(alpha), 'rgb%d' % S: np_(rgb), 'w%d' % S: np_(w),
                     'out%d' % S: np_(rgb_w), 'acc%d' % S: np_(acc), 'c1_%d' % S: np_(c1), 'c2_%d' % S: np_(c2),
                     'dalpha%d' % S: np_(alpha.grad), 'drgb%d' % S: np_(rgb.grad)})
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_composite.npz'), **comp)

    # --- march + secant, light visibility, unisurf, loss (full-width net)
    cfg = stage1_cfg('bunny')
    sd = stage1_state_dict(cfg, seed=11)
    rnet = rmdl.NeuralNetwork(cfg)
    rnet.load_state_dict(sd)
    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(sd)
    rren = rmdl.Renderer(rnet, cfg, device=torch.device('cpu'))
    oren = o1.Renderer(onet, cfg)
    h, w = 64, 80
    K, c2w, S_ = stage1_camera(cfg, h=h, w=w)
    g = torch.Generator().manual_seed(9)
    N = 96
    pix = torch.stack([torch.randint(0, w, (N,), generator=g).float(),
                       torch.randint(0, h, (N,), generator=g).float()], -1)[None]
    from model.common import origin_to_world, image_points_to_ray
    cam = origin_to_world(N, K, c2w, S_)
    rays = image_points_to_ray(pix, K, c2w)
    rays = rays / rays.norm(2, 2).unsqueeze(-1)
    with torch.no_grad():
        d_ref = rren.ray_marching(cam, rays, rnet, n_secant_steps=8, n_steps=[256, 257], rad=2.0,
                                  depth_range=[2, 6])
        d_ora = oren.ray_marching(cam, rays, n_steps=[256, 257], n_secant_steps=8, rad=2.0, depth_range=[2, 6])
    fin = torch.isfinite(d_ref)
    assert torch.equal(fin, torch.isfinite(d_ora))
    check('march d_i', d_ora[fin], d_ref[fin], 1e-6)
    print('  march: %d/%d rays hit' % (int((fin & (d_ref > 0)).sum()), N))

    surf = (cam + rays * torch.where(fin, d_ref, torch.ones_like(d_ref)).unsqueeze(-1))[0][fin[0] & (d_ref[0] > 0)][:48]
    ldir = torch.nn.functional.normalize(torch.randn(8, 3, generator=g), dim=-1)
    lv_ref = rren.light_visibility(surf=surf, light_dir=ldir)
    lv_ora = oren.light_visibility(surf=surf, light_dir=ldir)
    check('light_visibility', lv_ora, lv_ref, 5e-6)
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_march.npz'), sd_digest=state_dict_digest(sd),
                        pix=np_(pix), K=np_(K), c2w=np_(c2w), d_i=np_(d_ref), surf=np_(surf), ldir=np_(ldir),
                        light_vis=np_(lv_ref), hw=np.array([h, w]))

    batch = stage1_batch(cfg, h=h, w=w, seed=2)
    pix96 = pix
    g1 = torch.Generator().manual_seed(19)
    pix512 = torch.stack([torch.randint(0, w, (512,), generator=g1).float(), torch.randint(0, h, (512,), generator=g1).float()], -1)[None]
    # 'cfg1' = SURVEY 8(c) G6 at the scale of BASELINE configs[0]: bunny.yaml, 512 rays x 64 samples (it = 0), 256 march steps
    for it, pix, tag in ((0, pix96, 'it0'), (6000, pix96, 'it6000'), (0, pix512, 'cfg1')):
        N = pix.shape[1]
        for net in (rnet, onet):
            net.zero_grad()
        seed = 100 + it + (7 if tag == 'cfg1' else 0)
        torch.manual_seed(seed)
        out_r = rren(pix, K, c2w, S_, 'unisurf', add_noise=True, eval_=False, it=it)
        # capture the draws by replaying the RNG stream in the reference's order
        mask = out_r['mask_pred']
        n_hit = int(mask.sum())
        full = out_r['acc_map'].shape[1] and None
        torch.manual_seed(seed)
        torch.randint(256, 257, (1,))
        S = 96 if it > 5000 else 64
        nz_miss = torch.rand(1, N - n_hit, S)
        nz_hit = torch.rand(1, n_hit, S)
        nz_nbr = torch.rand(n_hit, 3)
        noise = {'miss': nz_miss, 'hit': nz_hit, 'nbr': nz_nbr}
        out_o = oren(pix, K, c2w, S_, 'unisurf', add_noise=True, eval_=False, it=it, noise=noise)
        assert torch.equal(out_o['mask_pred'], mask)
        for k in ('rgb', 'normal_pred', 'acc_map'):
            # (512 rays: the worst of 512 normalised gradients sits at 5e-6 -- different row blocking of the same GEMMs)
            check('unisurf %s %s' % (tag, k), out_o[k], out_r[k], 2e-6 if tag != 'cfg1' else 1e-5)
        # diff_norm = |n - n'| of two nearly equal unit normals: cancellation, so the
        # meaningful scale is the normals' (1.0), not max(diff_norm)
        dn_err = float((out_o['diff_norm'] - out_r['diff_norm']).abs().max())
        print('  [%s] oracle-vs-reference unisurf it=%d diff_norm abs=%.2e (max %.3e)' % (
            'ok ' if dn_err < 2e-6 else 'BAD', it, dn_err, float(out_r['diff_norm'].max())))
        assert dn_err < 2e-6
        # loss (losses.py:30-70) with synthetic GT
        rgb_gt = o1.gather_pixels(batch['img'], pix)
        ngt = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=torch.Generator().manual_seed(4)), dim=-1)
        nmask = torch.rand(1, N, generator=torch.Generator().manual_seed(5)) > 0.3
        rl = RLoss(1.0, 0.005, 0.05, 1.0, device=torch.device('cpu'))
        ol = o1.Loss(1.0, 0.005, 0.05, 1.0)
        tr = rl(out_r, rgb_gt, ngt, nmask)
        to = ol(out_o, rgb_gt, ngt, nmask)
        for k in tr:
            check('loss it=%d %s' % (it, k), to[k], tr[k], 2e-5 if k == 'grad_loss' else 2e-6)  # grad_loss = mean(diff_norm): cancellation
        tr['loss'].backward()
        to['loss'].backward()
        names, norms, projs = grad_digest({k: v.grad for k, v in rnet.named_parameters()})
        _, onorms, oprojs = grad_digest({k: v.grad for k, v in onet.named_parameters()})
        check('unisurf it=%d grad norms' % it, onorms, norms, 2e-5)
        check('unisurf it=%d grad projs' % it, oprojs, projs, 1e-4)
        np.savez_compressed(
            os.path.join(GOLDEN, 'stage1_unisurf_%s.npz' % tag), sd_digest=state_dict_digest(sd),
            pix=np_(pix), K=np_(K), c2w=np_(c2w), hw=np.array([h, w]), it=it,
            nz_miss=np_(nz_miss), nz_hit=np_(nz_hit), nz_nbr=np_(nz_nbr),
            rgb=np_(out_r['rgb']), mask_pred=np_(mask), diff_norm=np_(out_r['diff_norm']),
            normal_pred=np_(out_r['normal_pred']), acc_map=np_(out_r['acc_map']),
            rgb_gt=np_(rgb_gt), normal_gt=np_(ngt), norm_mask=np_(nmask),
            loss_names=np.array(sorted(tr.keys())), loss_vals=np.array([float(tr[k]) for k in sorted(tr.keys())]),
            grad_names=np.array(names), grad_norms=norms, grad_projs=projs)
    pix, N = pix96, 96
    # --- Renderer.shape_extract (rendering.py:297-376) as stage1/shape_extract.py:112-139 calls it: a chunk of the INT64
    #     x-major arange_pixels grid, 512 march steps, normals with tflag=False, shadow-ray visibility per 96-light batch
    from model.common import arange_pixels as r_arange
    hs, ws = 40, 48
    Ks, c2ws, Ss = stage1_camera(cfg, h=hs, w=ws)
    p_loc = r_arange(resolution=(hs, ws))[0]
    assert p_loc.dtype == torch.int64 and torch.equal(p_loc, o1.arange_pixels((hs, ws)))
    chunk = p_loc[:, 16 * hs + 8: 16 * hs + 8 + 200]  # 200 pixels out of the x-major walk (5 image columns)
    ldir_s = torch.nn.functional.normalize(torch.randn(5, 3, generator=torch.Generator().manual_seed(12)), dim=-1)
    with torch.no_grad():
        se_r = rren(chunk, Ks, c2ws, Ss, 'shape_extract', add_noise=False, eval_=True, it=100000, visibility=True, light_dir=ldir_s)
    se_o = oren(chunk, Ks, c2ws, Ss, 'shape_extract', visibility=True, light_dir=ldir_s)
    assert torch.equal(se_r['mask'], se_o['mask']) and 10 < int(se_r['mask'].sum()) < 200
    for k in ('normal', 'points', 'visibility'):
        check('shape_extract %s' % k, se_o[k], se_r[k], 5e-6)
    # --- Renderer.phong_renderer (rendering.py:228-293): the shaded preview of training.py:62-118, the third value of
    #     rendering_technique -- same 200-pixel chunk, the reference's own method
    with torch.no_grad():
        ph_r = rren(chunk.float(), Ks, c2ws, Ss, 'phong_renderer')
    ph_o = oren(chunk.float(), Ks, c2ws, Ss, 'phong_renderer')
    assert 10 < int((ph_r['rgb'] < 1).all(-1).sum()) < 200
    check('phong rgb', ph_o['rgb'], ph_r['rgb'], 5e-6)
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_phong.npz'), sd_digest=state_dict_digest(sd), hw=np.array([hs, ws]),
                        pix=np_(chunk), K=np_(Ks), c2w=np_(c2ws), rgb=np_(ph_r['rgb']))
    # --- Trainer.render_visdata (training.py:62-118): the periodic image grid of stage1/train.py -- the reference's OWN method on two
    #     small synthetic items (unisurf eval + phong preview over whole images in its 1024-pixel chunks, PIL output read back)
    import tempfile
    from PIL import Image
    hv_, wv_ = 20, 24
    vis_items = [stage1_batch(cfg, h=hv_, w=wv_, seed=s_) for s_ in (31, 32)]
    rtr_v = rmdl.Trainer(rren, None, cfg, device=torch.device('cpu'))
    png = os.path.join(tempfile.mkdtemp(), 'vis.png')
    rtr_v.render_visdata(vis_items, 1500, png)
    grid_ref = np.array(Image.open(png))
    assert grid_ref.shape == (2 * hv_, 8 * wv_, 3) and grid_ref.dtype == np.uint8   # 8 panels per item (a normal map is present)
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_visdata.npz'), sd_digest=state_dict_digest(sd), hw=np.array([hv_, wv_]),
                        seeds=np.array([31, 32]), it=1500, grid=grid_ref)
    rnet.train(); onet.train()
    rnet.train(); onet.train()  # shape_extract leaves the model in eval mode (rendering.py:311); no effect on these modules
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_shape_extract.npz'), sd_digest=state_dict_digest(sd), hw=np.array([hs, ws]),
                        pix=np_(chunk), K=np_(Ks), c2w=np_(c2ws), ldir=np_(ldir_s), mask=np_(se_r['mask']),
                        normal=np_(se_r['normal']), points=np_(se_r['points']), visibility=np_(se_r['visibility']))

    # --- Trainer.compute_loss (training.py:141-198): the reference's OWN trainer on a synthetic data dict; the pixel draw
    #     (common.py:32-35) and the renderer's draws are captured by replaying the RNG stream in the reference's order
    cl = {}
    hb, wb = 48, 64
    for tag, over, it, eval_mode in (('train', {}, 1500, False), ('eval', {}, 1500, True),
                                     ('mask', {'training.mask_loss': True, 'training.normal_after': 2000}, 1500, False)):
        cfg_t = stage1_cfg('bunny', **dict({'training.n_training_points': 160}, **over))
        data = stage1_batch(cfg_t, h=hb, w=wb, seed=6)
        data['img.mask_valid'] = (torch.rand(1, hb, wb, generator=torch.Generator().manual_seed(13)) > 0.1).float()
        if tag == 'mask':
            # BCE(acc, mask) with a mask that CONTRADICTS the geometry (acc ~ 1 where mask = 0) is -log(1 - acc) at the edge
            # of fp32 (gradient 1 / (1 - acc)): the reference's own value is then an accident of its summation order.  The
            # silhouette of the shape itself (dry march over the whole image, stored in the fixture) is what real masks are.
            ys, xs = torch.meshgrid(torch.arange(hb), torch.arange(wb), indexing='ij')
            allpix = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).float()[None]
            onet.load_state_dict(sd)
            with torch.no_grad():
                sil = torch.cat([o1.Renderer(onet, cfg_t)(c, data['img.camera_mat'], data['img.world_mat'], data['img.scale_mat'],
                                                          'unisurf', add_noise=False, eval_=True, it=it)['mask_pred']
                                 for c in torch.split(allpix, 1024, dim=1)])
            data['img.mask'] = sil.reshape(1, hb, wb).float()
            cl['mask_img'] = np_(sil.reshape(hb, wb))
        res = {}
        for who, net, Ren, Tr in (('ref', rnet, lambda n: rmdl.Renderer(n, cfg_t, device=torch.device('cpu')), None),
                                  ('ora', onet, lambda n: o1.Renderer(n, cfg_t), None)):
            net.load_state_dict(sd)
            net.zero_grad()
            ren_t = Ren(net)
            seed = 300 + it + int(eval_mode)
            if who == 'ref':
                trn = rmdl.Trainer(ren_t, torch.optim.Adam(net.parameters(), lr=1e-4), cfg_t, device=torch.device('cpu'))
                torch.manual_seed(seed)
                terms = trn.compute_loss(data, eval_mode=eval_mode, it=it)
                # replay the stream: px, py (common.py:32-35), randint (rendering.py:441), rand miss / hit (:139, :163),
                # rand_like neighbours (:204, training forward only)
                torch.manual_seed(seed)
                n = 160
                px = torch.randint(0, wb, size=(1, n, 1)).float()
                py = torch.randint(0, hb, size=(1, n, 1)).float()
                pix_t = torch.cat([px, py], dim=-1)
                torch.randint(256, 257, (1,))
                with torch.no_grad():
                    dry = o1.Renderer(onet, cfg_t)(pix_t, data['img.camera_mat'], data['img.world_mat'], data['img.scale_mat'],
                                                   'unisurf', add_noise=False, eval_=True, it=it)
                n_hit = int(dry['mask_pred'].sum())
                # the dry run consumed one randint: restore the stream position behind the reference's randint
                torch.manual_seed(seed)
                torch.randint(0, wb, size=(1, n, 1)); torch.randint(0, hb, size=(1, n, 1)); torch.randint(256, 257, (1,))
                S_t = 64
                noise_t = {'miss': torch.rand(1, n - n_hit, S_t), 'hit': torch.rand(1, n_hit, S_t)}
                if not eval_mode:
                    noise_t['nbr'] = torch.rand(n_hit, 3)
                assert 10 < n_hit < n - 10
            else:
                trn = o1.Trainer(ren_t, torch.optim.Adam(net.parameters(), lr=1e-4), cfg_t)
                terms = trn.compute_loss(data, eval_mode=eval_mode, it=it, pix=pix_t, noise=noise_t)
            if terms['loss'].requires_grad:
                terms['loss'].backward()
            res[who] = (terms, grad_digest({k: v.grad for k, v in net.named_parameters()}))
        (tr_, (names, norms, projs)), (to_, (_, onorms, oprojs)) = res['ref'], res['ora']
        assert sorted(tr_.keys()) == sorted(to_.keys()), (sorted(tr_.keys()), sorted(to_.keys()))
        for k in tr_:
            # grad_loss = mean |n - n'| (cancellation); mask_loss = BCE(acc): -log(1 - acc) near acc = 1 amplifies the 5e-7
            # weight_norm residue of acc by 1 / (1 - acc)
            check('compute_loss[%s] %s' % (tag, k), to_[k], tr_[k],
                  2e-5 if (k in ('grad_loss', 'mask_loss') or (k == 'loss' and 'mask_loss' in tr_)) else 2e-6)
        check('compute_loss[%s] grad norms' % tag, onorms, norms, 2e-5)
        check('compute_loss[%s] grad projs' % tag, oprojs, projs, 1e-4)
        lk = sorted(tr_.keys())
        cl.update({tag + '_pix': np_(pix_t), tag + '_it': it, tag + '_eval': eval_mode, tag + '_loss_names': np.array(lk),
                   tag + '_loss_vals': np.array([float(tr_[k]) for k in lk]), tag + '_grad_names': np.array(names),
                   tag + '_grad_norms': norms, tag + '_grad_projs': projs, tag + '_over': np.array(sorted(over.items()), dtype=object) if False else np.array([str(sorted(over.items()))])})
        cl.update({tag + '_nz_' + k: np_(v) for k, v in noise_t.items()})
    # the full-image branch (training.py:159-165): the reference itself cannot complete it -- record HOW it fails
    cfg_f = stage1_cfg('bunny', **{'training.n_training_points': 6 * 8})
    data_f = stage1_batch(cfg_f, h=6, w=8, seed=6)
    errs = []
    for trn in (rmdl.Trainer(rmdl.Renderer(rnet, cfg_f, device=torch.device('cpu')), None, cfg_f, device=torch.device('cpu')),
                o1.Trainer(o1.Renderer(onet, cfg_f), None, cfg_f)):
        try:
            trn.compute_loss(data_f, it=0)
            errs.append('')
        except Exception as e:  # noqa: BLE001
            errs.append('%s: %s' % (type(e).__name__, e))
    print('  full-image branch: reference -> %r' % errs[0])
    assert errs[0] == errs[1] and errs[0].startswith('RuntimeError: expected scalar type Float but found Long'), errs
    cl['full_image_error'] = np.array([errs[0]])
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_compute_loss.npz'), sd_digest=state_dict_digest(sd), hw=np.array([hb, wb]),
                        batch_seed=6, mask_valid_seed=13, n_points=160, **cl)
    # --- Trainer.train_step (training.py:46-60): two optimisation steps of the reference's OWN trainer (Adam, lr 1e-4), the
    #     draws of each step replayed in the reference's order; the hit count that sizes the tables comes from a dry march
    #     with the weights the step starts from
    cfg_s = stage1_cfg('bunny', **{'training.n_training_points': 160})
    data_s = stage1_batch(cfg_s, h=hb, w=wb, seed=8)
    rnet.load_state_dict(sd)
    onet.load_state_dict(sd)
    rtr = rmdl.Trainer(rmdl.Renderer(rnet, cfg_s, device=torch.device('cpu')), torch.optim.Adam(rnet.parameters(), lr=1e-4), cfg_s,
                       device=torch.device('cpu'))
    otr = o1.Trainer(o1.Renderer(onet, cfg_s), torch.optim.Adam(onet.parameters(), lr=1e-4), cfg_s)
    ts = {}
    for j, it in enumerate((1500, 1501)):
        seed = 700 + j
        n = 160
        torch.manual_seed(seed)
        px = torch.randint(0, wb, size=(1, n, 1)).float()
        py = torch.randint(0, hb, size=(1, n, 1)).float()
        pix_s = torch.cat([px, py], dim=-1)
        with torch.no_grad():
            dry = o1.Renderer(onet, cfg_s)(pix_s, data_s['img.camera_mat'], data_s['img.world_mat'], data_s['img.scale_mat'], 'unisurf',
                                           add_noise=False, eval_=True, it=it)
        n_hit = int(dry['mask_pred'].sum())
        torch.manual_seed(seed)
        tr_ = rtr.train_step(data_s, it=it)
        torch.manual_seed(seed)
        torch.randint(0, wb, size=(1, n, 1)); torch.randint(0, hb, size=(1, n, 1)); torch.randint(256, 257, (1,))
        noise_s = {'miss': torch.rand(1, n - n_hit, 64), 'hit': torch.rand(1, n_hit, 64), 'nbr': torch.rand(n_hit, 3)}
        to_ = otr.train_step(data_s, it=it, pix=pix_s, noise=noise_s)
        assert sorted(tr_) == sorted(to_)
        for k in tr_:
            check('train_step %d %s' % (j, k), to_[k], tr_[k], (2e-5 if k == 'grad_loss' else 2e-6) * (1 if j == 0 else 50))  # step 2 starts from Adam-updated weights
        lk = sorted(tr_)
        ts.update({'s%d_pix' % j: np_(pix_s), 's%d_it' % j: it, 's%d_loss_names' % j: np.array(lk),
                   's%d_loss_vals' % j: np.array([float(tr_[k]) for k in lk])})
        ts.update({'s%d_nz_%s' % (j, k): np_(v) for k, v in noise_s.items()})
    rsd2, osd2 = rnet.state_dict(), onet.state_dict()
    for k in rsd2:
        d = (rsd2[k] - osd2[k]).abs()
        assert float(d.max()) <= 2 * 2 * 1e-4 + 1e-6 and float(d.mean()) <= 1e-5, (k, float(d.max()), float(d.mean()))
    assert float((rsd2['lin0.weight_v'] - sd['lin0.weight_v']).abs().max()) > 0
    np.savez_compressed(os.path.join(GOLDEN, 'stage1_train_step.npz'), sd_digest=state_dict_digest(sd), hw=np.array([hb, wb]), batch_seed=8,
                        n_points=160, param_names=np.array(sorted(rsd2)), param_norms=grad_digest(rsd2)[1],
                        **ts, **{('p_' + k): np_(v.reshape(-1)[:1024]) for k, v in rsd2.items()})
    print('stage1 goldens written')


# ---------------------------------------------------------------------------
def gen_stage2():
    import types
    from oracle import stage2 as o2
    # harness-side shims (no reference edits): the REAL utils/rend_util.py is imported; only the third-party modules it
    # needs at import time and that this image lacks (imageio incl. its download-at-import, skimage, cv2) are empty
    # placeholders in sys.modules -- get_camera_params / lift (rend_util.py:90-147) use none of them.
    for name in ('imageio', 'imageio.plugins', 'imageio.plugins.freeimage', 'skimage', 'cv2'):  # (eval_utils needs imageio + cv2 too)
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['imageio'].plugins = sys.modules['imageio.plugins']
    sys.modules['imageio.plugins'].freeimage = sys.modules['imageio.plugins.freeimage']
    sys.modules['imageio.plugins.freeimage'].download = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self  # reference code hard-codes .cuda()
    sys.path.insert(0, os.path.join(REF, 'stage2'))
    from utils import rend_util as RREND  # the reference's own module
    assert os.path.realpath(RREND.__file__).startswith(os.path.realpath(REF)), RREND.__file__
    from model.renderer import PSNetwork as RPS
    from model.loss import MainLoss as RMain, NormalLoss as RNormal
    from model.sgbasis import SGBasis as RSG
    from model.microfacet import Microfacet as RMF

    # ---- camera rays: the reference's get_camera_params vs the oracle's restatement, + a fixture for the HIP side
    from psnerf_amd.synthetic import look_at_pose
    gc = torch.Generator().manual_seed(8)
    uv_c = torch.stack([torch.randint(0, 612, (300,), generator=gc).float(), torch.randint(0, 512, (300,), generator=gc).float()], -1)[None]
    K_c = torch.eye(4)[None].clone()
    K_c[0, 0, 0], K_c[0, 1, 1], K_c[0, 0, 2], K_c[0, 1, 2] = 3759.0, 3741.5, 306.0, 256.0  # fx != fy: both are used (rend_util.py:131-139)
    pose_c = look_at_pose(31.5, az_deg=40.0, el_deg=-15.0)[None]
    rd_r, loc_r = RREND.get_camera_params(uv_c, pose_c, K_c)
    rd_o, loc_o = o2.camera_rays(uv_c, pose_c, K_c)
    check('camera rays', rd_o, rd_r, 1e-7)
    check('camera loc', loc_o, loc_r, 0.0)
    np.savez_compressed(os.path.join(GOLDEN, 'stage2_camera.npz'), uv=np_(uv_c), K=np_(K_c), pose=np_(pose_c),
                        ray_dirs=np_(rd_r), cam_loc=np_(loc_r))

    # ---- environment-map light grid + pixel split / merge: the reference's own eval_utils / general modules
    from utils import eval_utils as REVAL
    from utils import general as RGEN
    from psnerf_amd.stage2 import relight as prl
    assert os.path.realpath(REVAL.__file__).startswith(os.path.realpath(REF))
    xyz_r, areas_r = REVAL.gen_light_xyz(16, 32, envmap_radius=1)   # eval.py:203 uses the default radius; direction is normalised after
    xyz_p, areas_p = prl.gen_light_xyz(16, 32, envmap_radius=1)
    check('gen_light_xyz', xyz_p, xyz_r, 1e-15)
    check('gen_light_xyz areas', areas_p, areas_r, 1e-15)
    npx = 2500  # the reference splits into chunks of 1024 pixels (general.py:28)
    mi = {'uv': torch.rand(1, npx, 2), 'points': torch.rand(1, npx, 3), 'intrinsics': torch.eye(4)[None],
          'object_mask': torch.rand(1, npx) > 0.5}
    sp_r = RGEN.split_input(dict(mi), npx)
    sp_p = prl.split_input(dict(mi), npx, n_pixels=1024)
    assert len(sp_r) == len(sp_p) == 3
    for a, b in zip(sp_r, sp_p):
        for k in ('uv', 'points', 'object_mask'):
            assert torch.equal(a[k], b[k]), k
    res = [{'sg_rgb_values': torch.rand(2, 1024, 3), 'mask': torch.rand(1, 1024) > 0.5} for _ in range(2)] + \
          [{'sg_rgb_values': torch.rand(2, 452, 3), 'mask': torch.rand(1, 452) > 0.5}]
    mr, mp = RGEN.merge_output(res, npx, 1), prl.merge_output(res, npx, 1)
    for k in mr:
        assert torch.equal(mr[k], mp[k]), k
    np.savez_compressed(os.path.join(GOLDEN, 'stage2_light_xyz.npz'), xyz=xyz_r, areas=areas_r)
    print('  [ok ] relight helpers == reference eval_utils.gen_light_xyz / general.split_input / merge_output')

    # ---- environment-map relighting (stage2/eval.py:173-218): the loop of evaluate() over light batches and pixel chunks,
    #      re-assembled from the reference's OWN pieces -- PSNetwork, eval_utils.gen_light_xyz, general.split_input /
    #      merge_output -- in the order eval.py runs them (evaluate() itself is one function around datasets and checkpoints)
    conf_r = o2.bear_conf()
    sd_r = stage2_state_dict(conf_r, seed=12)
    rnet_r = RPS(conf_r)
    rnet_r.load_state_dict(sd_r)
    rnet_r.eval()
    hr, wr, lh_r, lbatch = 30, 40, 4, 10
    inp_r, _ = stage2_inputs(hr * wr, 1, 1, seed=3, h=hr, w=wr)
    uv_r = np.mgrid[0:hr, 0:wr].astype(np.int32)                      # eval.py:178-180
    uv_r = torch.from_numpy(np.flip(uv_r, axis=0).copy()).float().reshape(2, -1).transpose(1, 0)
    mi_r = {'object_mask': torch.ones(1, hr * wr), 'uv': uv_r[None], 'intrinsics': inp_r['intrinsics'], 'pose': inp_r['pose'],
            'normal': torch.ones(1, hr * wr, 3), 'points': inp_r['points'], 'surface_mask': inp_r['surface_mask']}
    env_r = np.random.RandomState(5).rand(lh_r, 2 * lh_r, 3).astype(np.float32) * 0.3
    lxyz_r = REVAL.gen_light_xyz(lh_r, 2 * lh_r, envmap_radius=1)[0].reshape(-1, 3)
    envf = env_r.reshape(-1, 3)
    rgb_all, vis_all = [], []
    with torch.no_grad():
        for lstart in range(0, lh_r ** 2 * 2, lbatch):                # eval.py:196-213
            lend = min(lh_r ** 2 * 2, lstart + lbatch)
            mi_r['light_direction'] = torch.nn.functional.normalize(torch.tensor(lxyz_r[lstart:lend]).float(), p=2, dim=-1)
            mi_r['light_intensity'] = torch.tensor(envf[lstart:lend]).float()
            res = []
            for sp in RGEN.split_input(mi_r, hr * wr):
                out = rnet_r(sp)
                res.append({'sg_rgb_values': out['sg_rgb_values'].detach(), 'visibility': out.get('visibility', torch.ones_like(out['points'])).detach()})
            mo = RGEN.merge_output(res, hr * wr, 1)
            rgb_all.append(np_(mo['sg_rgb_values'].reshape(-1, hr, wr, 3)))
            vis_all.append(np_(mo['visibility'].reshape(-1, hr, wr, 3)))
    rgb_ref = np.concatenate(rgb_all, 0).sum(0).clip(0, 1)             # eval.py:214
    vis_ref = np.concatenate(vis_all, 0).mean(0)                       # eval.py:220
    onet_r = o2.PSNetwork(conf_r)
    onet_r.load_state_dict(sd_r)
    with torch.no_grad():
        mi_o = {k: v for k, v in mi_r.items() if k not in ('light_direction', 'light_intensity')}
        mi_o['object_mask'] = mi_o['object_mask'].bool()
        mi_o['light_direction'] = torch.nn.functional.normalize(torch.tensor(lxyz_r).float(), p=2, dim=-1)
        mi_o['light_intensity'] = torch.tensor(envf).float()
        oo = onet_r(mi_o)
    check('relight rgb', oo['sg_rgb_values'].sum(0).clamp(0, 1).reshape(hr, wr, 3), rgb_ref, 2e-6)
    check('relight visibility', oo['visibility'].mean(0).reshape(hr, wr, 3), vis_ref, 2e-6)
    np.savez_compressed(os.path.join(GOLDEN, 'stage2_relight.npz'), sd_digest=state_dict_digest(sd_r), hw=np.array([hr, wr]), light_h=lh_r,
                        input_seed=3, uv=np_(uv_r), env=env_r, rgb=rgb_ref.astype(np.float32), visibility=vis_ref.astype(np.float32))

    # ---- test-view render of evaluate() (stage2/eval.py:314-417) and its material-edit variant (:233-312), re-assembled from the
    #      reference's OWN pieces in the order eval.py runs them: PSNetwork (eval mode, jitter 0 as :47-48 sets it), the optimised light
    #      tables of a model trained with train.light_train on all views (:338-345), general.split_input / merge_output per light batch
    conf_v = o2.bear_conf(**{'brdf.net.xyz_jitter_std': 0, 'normal.net.xyz_jitter_std': 0})
    sd_v = stage2_state_dict(conf_v, seed=14)
    rnet_v = RPS(conf_v)
    rnet_v.load_state_dict(sd_v)
    rnet_v.eval()
    hv, wv, Lv, lbatch_v, NLv, off_v = 36, 34, 5, 2, 9, 3          # 1224 pixels = two chunks of general.py's 1024
    inp_v, _ = stage2_inputs(hv * wv, 1, 1, seed=4, h=hv, w=wv)
    uv_v = np.mgrid[0:hv, 0:wv].astype(np.int32)                    # eval.py:320-322
    uv_v = torch.from_numpy(np.flip(uv_v, axis=0).copy()).float().reshape(2, -1).transpose(1, 0)
    gv = torch.Generator().manual_seed(15)
    light_para_v = torch.nn.Embedding(NLv, 3, sparse=True).eval().requires_grad_(False)
    light_para_v.weight.data.copy_(torch.nn.functional.normalize(inp_v['pose'][0, :3, 3][None] + 8.0 * torch.randn(NLv, 3, generator=gv), dim=-1) * 1.7)
    light_inten_v = torch.nn.Embedding(NLv, 1, sparse=True).eval().requires_grad_(False)
    light_inten_v.weight.data.copy_(1.5 + torch.rand(NLv, 1, generator=gv))
    lidx_v = torch.arange(Lv).long()

    def eval_view(albedo_new=None, basis_new=None):
        mi_v = {'object_mask': torch.ones(1, hv * wv), 'uv': uv_v[None], 'intrinsics': inp_v['intrinsics'], 'lidx': lidx_v, 'pose': inp_v['pose'],
                'normal': torch.ones(1, hv * wv, 3), 'points': inp_v['points'], 'surface_mask': inp_v['surface_mask']}
        rgb_all, vis_all, rough_all = [], [], []
        with torch.no_grad():
            for lstart in range(0, Lv, lbatch_v):                   # eval.py:339-363
                lend = min(Lv, lstart + lbatch_v)
                l_slt = off_v + lidx_v[lstart:lend]
                mi_v['light_direction'] = torch.nn.functional.normalize(light_para_v(l_slt), p=2, dim=-1)
                mi_v['light_intensity'] = light_inten_v(l_slt)
                res = []
                for sp in RGEN.split_input(mi_v, hv * wv):
                    out = rnet_v(sp, albedo_new=albedo_new, basis_new=basis_new) if (albedo_new is not None or basis_new is not None) else rnet_v(sp)
                    res.append({k: out[k].detach() for k in out})
                mo = RGEN.merge_output(res, hv * wv, 1)
                rgb_all.append(np_(mo['sg_rgb_values'].reshape(-1, hv, wv, 3)))
                rough_all.append(np_(mo['sg_specular_rgb_values'].reshape(-1, hv, wv, 3)))
                vis_all.append(np_(mo['visibility'].reshape(-1, hv, wv, 3)))
        rmask = np_(mo['network_object_mask'].reshape(hv, wv))      # :382
        return {'rgb': np.concatenate(rgb_all, 0).clip(0, 1), 'rough': np.concatenate(rough_all, 0), 'mask': rmask,
                'normal': np_(mo['normal_pred'].reshape(hv, wv, 3)) * rmask[..., None],          # :394-395
                'albedo': np_(mo['sg_diffuse_albedo_values'].reshape(hv, wv, 3)).clip(0, 1),      # :400
                'visibility': np.concatenate(vis_all, 0).clip(0, 1)}                              # :406
    ref_v = eval_view()
    color_v, basis_v = '#4080c0', 3
    albedo_new_v = (np.array([int(color_v.lstrip('#')[i:i + 2], 16) for i in (0, 2, 4)]).astype(np.float32) / 5. / 255.).astype(np.float32)  # :127-130
    ref_e = eval_view(albedo_new=albedo_new_v, basis_new=basis_v)
    from psnerf_amd.stage2 import relight as prl2
    an, bn, name = prl2.edit_material(color=color_v, basis=basis_v, edit_albedo=True, edit_specular=True)
    assert np.array_equal(an, albedo_new_v) and bn == basis_v and name == '#4080c0_sg4'
    # the oracle network under the product's loop (relight.render_view) == the reference's pieces
    onet_v = o2.PSNetwork(conf_v)
    onet_v.load_state_dict(sd_v)
    onet_v.eval()
    mi_o = {'object_mask': torch.ones(1, hv * wv, dtype=torch.bool), 'uv': uv_v[None], 'intrinsics': inp_v['intrinsics'], 'pose': inp_v['pose'],
            'normal': torch.ones(1, hv * wv, 3), 'points': inp_v['points'], 'surface_mask': inp_v['surface_mask']}
    ld_o, li_o = prl2.eval_lights(None, lidx_v, light_para_v, light_inten_v, light_offset=off_v)
    for tag, ref_maps, kw in (('view', ref_v, {}), ('edit', ref_e, {'albedo_new': albedo_new_v, 'basis_new': basis_v})):
        got = prl2.render_view(onet_v, mi_o, ld_o, li_o, light_batch=lbatch_v, pixel_chunk=1024, **kw)
        for k in ('rgb', 'rough', 'visibility'):
            check('eval %s %s' % (tag, k), got[k].reshape(-1, hv, wv, 3), ref_maps[k], 2e-6)
        for k in ('normal', 'albedo'):
            check('eval %s %s' % (tag, k), got[k].reshape(hv, wv, 3), ref_maps[k], 2e-6)
        assert np.array_equal(np_(got['mask']).reshape(hv, wv), ref_maps['mask'].astype(bool))
    np.savez_compressed(os.path.join(GOLDEN, 'stage2_eval_view.npz'), sd_digest=state_dict_digest(sd_v), hw=np.array([hv, wv]), input_seed=4,
                        uv=np_(uv_v), light_para=np_(light_para_v.weight), light_inten=np_(light_inten_v.weight), lidx=np_(lidx_v), light_offset=off_v,
                        light_batch=lbatch_v, color=color_v, basis=basis_v,
                        **{('view_' + k): v.astype(np.float32) for k, v in ref_v.items()}, **{('edit_' + k): v.astype(np.float32) for k, v in ref_e.items()})

    # ---- normal jitter > 0 (renderer.py:133-140; bear.conf has 0): TWO torch.normal draws, normal jitter first
    conf_j = o2.bear_conf(**{'normal.net.xyz_jitter_std': 0.02})
    sd_j = stage2_state_dict(conf_j, seed=35)
    rnet, onet = RPS(conf_j), o2.PSNetwork(conf_j)
    rnet.load_state_dict(sd_j)
    onet.load_state_dict(sd_j)
    N, L, V = 300, 4, 3
    inp, gt = stage2_inputs(N, L, V, seed=61)
    ns = int(inp['surface_mask'].sum())
    torch.manual_seed(79)
    nz_n = torch.normal(0, torch.ones(ns, 3) * 0.02)
    nz_x = torch.normal(0, torch.ones(ns, 3) * 0.01)
    res = []
    for net, Main, Norm, kw in ((rnet, RMain, RNormal, {}), (onet, o2.MainLoss, o2.NormalLoss, {'noise': {'normal': nz_n, 'xyz': nz_x}})):
        i2 = {k: v.clone() for k, v in inp.items()}
        ldir = i2['light_direction'].clone().requires_grad_(True)
        i2['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
        torch.manual_seed(79)
        out = net(i2, **kw)
        t = Main(loss_type='L1', sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)(out, gt, i2)
        tn = Norm(1, 0.05)(out)
        total = t['loss'] + tn['loss']
        total.backward()
        gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        res.append((out, dict(t, total=total, normal_loss=tn['normal_loss'], normal_smooth_loss=tn['normal_smooth_loss']), gr))
    assert 'normal_jitter' in res[0][0] and res[0][1]['normal_smooth_loss'] is not None
    keys = [k for k in res[0][0] if torch.is_tensor(res[0][0][k]) and res[0][0][k].dtype.is_floating_point]
    for k in keys:
        check('normal-jitter %s' % k, res[1][0][k], res[0][0][k], 2e-6)
    for k in res[0][1]:
        if res[0][1][k] is not None:
            check('normal-jitter loss %s' % k, res[1][1][k], res[0][1][k], 2e-6)
    names, norms, projs = grad_digest(res[0][2])
    _, onorms, oprojs = grad_digest(res[1][2])
    check('normal-jitter grad norms', onorms, norms, 2e-5)
    check('normal-jitter grad projs', oprojs, projs, 1e-4)
    lk = sorted(k for k in res[0][1] if res[0][1][k] is not None)
    np.savez_compressed(
        os.path.join(GOLDEN, 'stage2_psnet_normal_jitter.npz'), sd_digest=state_dict_digest(sd_j), N=N, L=L, V=V,
        input_seed=61, nz_normal=np_(nz_n), nz_xyz=np_(nz_x), loss_names=np.array(lk),
        loss_vals=np.array([float(res[0][1][k]) for k in lk]), grad_names=np.array(names), grad_norms=norms,
        grad_projs=projs, **{('out_' + k): np_(res[0][0][k]) for k in keys})

    g = torch.Generator().manual_seed(21)
    n = 200
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    nn_ = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    l = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    alb = torch.rand(n, 3, generator=g)
    wts = torch.rand(n, 27, generator=g)
    rb, rs = RSG(nbasis=9, specular_rgb=True)(v=v, n=nn_, l=l, albedo=alb, weights=wts)
    ob, os_ = o2.SGBasis(nbasis=9, specular_rgb=True)(v=v, n=nn_, l=l, albedo=alb, weights=wts)
    check('sgbasis brdf', ob, rb)
    rough = torch.rand(n, 1, generator=g) * 0.8 + 0.1
    l2 = torch.nn.functional.normalize(torch.randn(n, 5, 3, generator=g), dim=-1)
    mf_r = RMF(f0=0.05)(l2, v, nn_, albedo=alb, rough=rough)
    mf_o = o2.microfacet_brdf(l2, v, nn_, alb, rough, f0=0.05)
    check('microfacet', mf_o, mf_r, 1e-5)
    np.savez_compressed(os.path.join(GOLDEN, 'stage2_brdf.npz'), v=np_(v), n=np_(nn_), l=np_(l), albedo=np_(alb),
                        weights=np_(wts), brdf=np_(rb), spec=np_(rs), rough=np_(rough), l2=np_(l2), mf=np_(mf_r))

    # ---- GGX microfacet render model (train.render_model = microfacet), full PSNetwork + losses + grads
    conf_m = o2.bear_conf(**{'train.render_model': 'microfacet'})
    sd_m = stage2_state_dict(conf_m, seed=33)
    rnet, onet = RPS(conf_m), o2.PSNetwork(conf_m)
    rnet.load_state_dict(sd_m)
    onet.load_state_dict(sd_m)
    N, L, V = 384, 6, 4
    inp, gt = stage2_inputs(N, L, V, seed=52)
    ns = int(inp['surface_mask'].sum())
    torch.manual_seed(78)
    nz = torch.normal(0, torch.ones(ns, 3) * 0.01)
    res = []
    for net, Main, Norm, kw in ((rnet, RMain, RNormal, {}), (onet, o2.MainLoss, o2.NormalLoss, {'noise': {'xyz': nz}})):
        i2 = {k: v.clone() for k, v in inp.items()}
        ldir = i2['light_direction'].clone().requires_grad_(True)
        lint = i2['light_intensity'].clone().requires_grad_(True)
        i2['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
        i2['light_intensity'] = lint
        torch.manual_seed(78)
        out = net(i2, **kw)
        t = Main(loss_type='L1', sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)(out, gt, i2)
        tn = Norm(1, 0.05)(out)
        total = t['loss'] + tn['loss']
        total.backward()
        gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        gr['__light_dir'], gr['__light_int'] = ldir.grad, lint.grad
        res.append((out, dict(t, total=total, normal_loss=tn['normal_loss']), gr))
    keys = [k for k in res[0][0] if torch.is_tensor(res[0][0][k]) and res[0][0][k].dtype.is_floating_point]
    for k in keys:
        check('microfacet %s' % k, res[1][0][k], res[0][0][k], 2e-6)
    for k in res[0][1]:
        if res[0][1][k] is not None:
            check('microfacet loss %s' % k, res[1][1][k], res[0][1][k], 2e-6)
    names, norms, projs = grad_digest(res[0][2])
    _, onorms, oprojs = grad_digest(res[1][2])
    check('microfacet grad norms', onorms, norms, 2e-5)
    check('microfacet grad projs', oprojs, projs, 1e-4)
    lk = sorted(k for k in res[0][1] if res[0][1][k] is not None)
    np.savez_compressed(
        os.path.join(GOLDEN, 'stage2_psnet_microfacet.npz'), sd_digest=state_dict_digest(sd_m), N=N, L=L, V=V,
        input_seed=52, nz_xyz=np_(nz), loss_names=np.array(lk), loss_vals=np.array([float(res[0][1][k]) for k in lk]),
        grad_names=np.array(names), grad_norms=norms, grad_projs=projs,
        g_light_dir=np_(res[0][2]['__light_dir']), g_light_int=np_(res[0][2]['__light_int']),
        **{('out_' + k): np_(res[0][0][k]) for k in keys})

    conf = o2.bear_conf()
    sd = stage2_state_dict(conf, seed=31)
    for L in (1, 10):
        for phase in (1, 2):
            rnet, onet = RPS(conf), o2.PSNetwork(conf)
            rnet.load_state_dict(sd)
            onet.load_state_dict(sd)
            N, V = 512, 8
            inp, gt = stage2_inputs(N, L, V, seed=40 + L)
            if phase == 1:  # train_fix iters 0..4999 (trainer.py:485-500)
                lw = dict(sg_rgb_weight=0, albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=10)
                for net in (rnet, onet):
                    net.albedo_net.eval().requires_grad_(False)
                    net.rough_net.eval().requires_grad_(False)
            else:
                lw = dict(sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)
            ns = int(inp['surface_mask'].sum())
            seed = 77
            torch.manual_seed(seed)
            nz = torch.normal(0, torch.ones(ns, 3) * 0.01)
            outs, terms, grads = [], [], []
            for net, Main, Norm, kw in ((rnet, RMain, RNormal, {}), (onet, o2.MainLoss, o2.NormalLoss, {'noise': {'xyz': nz}})):
                i2 = {k: v.clone() for k, v in inp.items()}
                ldir = i2['light_direction'].clone().requires_grad_(phase == 2)
                lint = i2['light_intensity'].clone().requires_grad_(phase == 2)
                i2['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
                i2['light_intensity'] = lint
                torch.manual_seed(seed)
                out = net(i2, **kw)
                t = Main(loss_type='L1', **lw)(out, gt, i2)
                tn = Norm(1, 0.05)(out)
                total = t['loss'] + tn['loss']
                total.backward()
                gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
                if phase == 2:
                    gr['__light_dir'] = ldir.grad
                    gr['__light_int'] = lint.grad
                outs.append(out)
                t = dict(t)
                t['normal_loss'] = tn['normal_loss']
                t['total'] = total
                terms.append(t)
                grads.append(gr)
            keys = [k for k in outs[0] if torch.is_tensor(outs[0][k]) and outs[0][k].dtype.is_floating_point]
            for k in keys:
                check('L=%d ph=%d %s' % (L, phase, k), outs[1][k], outs[0][k], 2e-6)
            for k in terms[0]:
                if terms[0][k] is not None:
                    check('L=%d ph=%d loss %s' % (L, phase, k), terms[1][k], terms[0][k], 2e-6)
            names, norms, projs = grad_digest(grads[0])
            onames, onorms, oprojs = grad_digest(grads[1])
            assert names == onames, (names, onames)
            check('L=%d ph=%d grad norms' % (L, phase), onorms, norms, 2e-5)
            check('L=%d ph=%d grad projs' % (L, phase), oprojs, projs, 1e-4)
            save = {('out_' + k): np_(outs[0][k]) for k in keys}
            lk = sorted(k for k in terms[0] if terms[0][k] is not None)
            extra = {}
            if phase == 2:
                extra = {'g_light_dir': np_(grads[0]['__light_dir']), 'g_light_int': np_(grads[0]['__light_int'])}
            np.savez_compressed(
                os.path.join(GOLDEN, 'stage2_psnet_L%d_ph%d.npz' % (L, phase)), sd_digest=state_dict_digest(sd),
                N=N, L=L, V=V, input_seed=40 + L, nz_xyz=np_(nz), loss_names=np.array(lk),
                loss_vals=np.array([float(terms[0][k]) for k in lk]), grad_names=np.array(names),
                grad_norms=norms, grad_projs=projs, **save, **extra)
    print('stage2 goldens written')


# ---------------------------------------------------------------------------
TRAINER_VARIANTS = {   # fixture name -> the switches of stage2/trainer.py:36-50 that differ from bear.conf
    None: {},
    'gtlight': {'train.light_train': False},      # ground-truth lights: no tables, no SparseAdam, vis loss on the L shading rows
    'fixlight': {'train.ana_fixlight': True},     # the light tables stay frozen behind the iteration-5000 switch
    'novisloss': {'train.vis_loss': False, 'train.multi_light': False},  # visibility net frozen at iteration 0; single-light layout
}                                                  # (under multi_light the reference fails at trainer.py:366 with KeyError 'visibility')


def trainer_fixture_name(vis_plus=False, inten_train=True, variant=None):
    if variant is not None:
        return 'stage2_trainer_%s.npz' % variant
    return 'stage2_trainer_visplus.npz' if vis_plus else ('stage2_trainer.npz' if inten_train else 'stage2_trainer_nointen.npz')


def gen_trainer(vis_plus=False, inten_train=True, variant=None):
    """stage2/trainer.py: the reference's OWN TrainRunner.run / train_fix (the step body :355-410,462-464 and the schedule
    :485-513) driven for six iterations across the iteration-5000 switch.  The module imports with empty placeholders for
    the third-party packages this image lacks (pyhocon, tensorboardX, imageio, cv2, skimage, plotly, GPUtil, trimesh: none is
    touched by the step body; likewise torchvision, which utils/plots.py imports), and ``run`` is called on a duck-typed ``self`` that carries exactly the attributes the loop
    reads: a list as the data loader, the reference's model / loss classes, torch's Adam / SparseAdam as trainer.py:126-168
    constructs them.  No reference edits.  Writes tests/golden/stage2_trainer.npz (vis_plus: stage2_trainer_visplus.npz;
    inten_train=False -- train.light_inten_train absent, as in bunny.conf / armadillo.conf: no intensity table, the model shades with
    its scalar brdf.light_intensity, trainer.py:38,154-163,378-379 -- stage2_trainer_nointen.npz).  ``variant``: one of
    TRAINER_VARIANTS -- the trainer switches no shipped configuration uses (stage2_trainer_<variant>.npz)."""
    import tempfile
    import types
    from oracle import stage2 as o2
    for name in ('imageio', 'imageio.plugins', 'imageio.plugins.freeimage', 'skimage', 'skimage.measure', 'cv2', 'pyhocon', 'tensorboardX',
                 'plotly', 'plotly.graph_objs', 'plotly.offline', 'plotly.subplots', 'GPUtil', 'trimesh', 'torchvision'):
        if name not in sys.modules:
            try:
                __import__(name)  # whatever the image does have is used as it is
            except Exception:  # noqa: BLE001
                sys.modules[name] = types.ModuleType(name)
    sys.modules['pyhocon'].ConfigFactory = object
    sys.modules['tensorboardX'].SummaryWriter = object
    sys.modules['imageio'].plugins = sys.modules['imageio.plugins']
    sys.modules['imageio.plugins'].freeimage = sys.modules['imageio.plugins.freeimage']
    sys.modules['imageio.plugins.freeimage'].download = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, os.path.join(REF, 'stage2'))
    import trainer as RT
    assert os.path.realpath(RT.__file__).startswith(os.path.realpath(REF)), RT.__file__
    from model.renderer import PSNetwork as RPS
    from model.loss import MainLoss as RMain, NormalLoss as RNormal

    over = dict(TRAINER_VARIANTS[variant])
    multi_light = bool(over.get('train.multi_light', True))
    light_train = bool(over.get('train.light_train', True))
    conf = o2.bear_conf(**{'train.vis_train_num': 5, 'train.light_inten_train': bool(inten_train), **over})
    sd = stage2_state_dict(conf, seed=41)
    N, L, V, NL = 360, 4, 3, 12
    light_slt = [list(range(5)), list(range(7))]          # two views with 5 and 7 lights: rows 0..4 and 5..11 of the tables
    if not multi_light:   # one light per view (dataset.py:141-147): the tables have one row per view, l_slt = [view] (trainer.py:375)
        L, NL, light_slt = 1, 2, [[0], [0]]
    g = torch.Generator().manual_seed(17)
    light_init = [torch.nn.functional.normalize(torch.randn(len(ls), 3, generator=g), dim=-1) * 1.3 for ls in light_slt]  # un-normalised on purpose
    batches = []
    for b in range(3):
        inp, gt = stage2_inputs(N, L, V, seed=200 + b)
        view = b % 2
        lidx = torch.randperm(len(light_slt[view]), generator=g)[:L]
        mi = {k: inp[k] for k in ('intrinsics', 'uv', 'pose', 'object_mask', 'surface_mask', 'points', 'normal')}
        if multi_light:
            mi['light_direction'] = inp['light_direction'][None]          # collated: batch dimension (trainer.py:366 strips it)
            mi['visibility'] = inp['visibility'][None]
            mi['lidx'] = lidx[None]
            batches.append((torch.tensor([view]), mi, {'rgb': gt['rgb'][None]}))
        else:   # single-light items: light_direction [3] -> collated [1,3], rgb [N,3] -> [1,N,3]; no 'visibility' without vis_loss (dataset.py:168)
            mi['light_direction'] = inp['light_direction']
            batches.append((torch.tensor([view]), mi, {'rgb': gt['rgb']}))

    # train.vis_plus (trainer.py:209-214, 384-392): per view P extra directions with their stage-1 visibility maps over the WHOLE
    # view (hw pixels) + the view's own lights / visibility; every step draws vnum of the P + L_v rows with np.random.choice
    # and looks the sampled pixels up through model_input['sampling_idx']
    P, hw, vnum = 6, 900, 5
    vp_light = [torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1) for _ in light_slt]
    vp_vis = [(torch.rand(P, hw, generator=g) > 0.3).float() for _ in light_slt]
    view_vis = [(torch.rand(len(ls), hw, generator=g) > 0.3).float() for ls in light_slt]
    samp = [torch.randperm(hw, generator=g)[:N] for _ in range(3)]
    if vis_plus:
        for b, (idx, mi, gt) in enumerate(batches):
            view = int(idx[0])
            mi['vidx'], mi['vidx_ori'], mi['sampling_idx'] = torch.tensor([view]), torch.tensor([view]), samp[b][None]

    class Recorder(object):   # the reference's loss module, every call logged
        def __init__(self, inner):
            self.inner, self.log = inner, []
        def __getattr__(self, k):
            return getattr(self.__dict__['inner'], k)
        def __setattr__(self, k, v):
            if k in ('inner', 'log'):
                self.__dict__[k] = v
            else:
                setattr(self.__dict__['inner'], k, v)
        def __call__(self, *a, **k):
            out = self.inner(*a, **k)
            self.log.append({kk: (float(vv.detach()) if vv is not None else None) for kk, vv in out.items()})
            return out

    rnet = RPS(conf)
    rnet.load_state_dict(sd)
    ns = types.SimpleNamespace()
    ns.conf, ns.device = conf, torch.device('cpu')
    ns.model = rnet
    ns.loss = Recorder(RMain(loss_type='L1', sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1))
    ns.loss_n = Recorder(RNormal(1, 0.05))
    ns.normal_train, ns.multi_light, ns.light_train, ns.light_inten_train = True, multi_light, light_train, bool(inten_train) and light_train
    ns.visibility, ns.vis_loss, ns.vis_plus, ns.ana_fixlight, ns.light_decay, ns.train_order = \
        True, bool(over.get('train.vis_loss', True)), bool(vis_plus), bool(over.get('train.ana_fixlight', False)), False, True
    ns.vis_plus_light = {'view_%02d' % (v + 1): vp_light[v].numpy().tolist() for v in range(2)}      # vis_plus/light_dir.json
    ns.vis_plus_all = {'view_%02d' % (v + 1): vp_vis[v].numpy().reshape(P, 30, 30) for v in range(2)}   # vis_plus/view_XX.npy
    # learning rates of stage2/confs/bear.conf:19-20,48-50 (the milestones lie beyond these iterations)
    ns.sg_optimizer = torch.optim.Adam(rnet.parameters(), lr=5e-4)
    ns.sg_scheduler = torch.optim.lr_scheduler.MultiStepLR(ns.sg_optimizer, [], gamma=0.5)
    if not multi_light:
        light_init = [li[:1] for li in light_init]
    if light_train:   # (trainer.py:126-168; without it the runner has none of these attributes and the loop must never touch them)
        ns.light_para = torch.nn.Embedding(NL, 3, sparse=True)
        ns.light_para.weight.data.copy_(torch.cat(light_init, dim=0))
        ns.light_vis_train = [li.clone() for li in light_init]
    if not light_train:
        pass
    elif inten_train:
        ns.light_inten_para = torch.nn.Embedding(NL, 1, sparse=True)
        torch.nn.init.constant_(ns.light_inten_para.weight, rnet.light_int)
        ns.light_optimizer = torch.optim.SparseAdam(
            [{'params': list(ns.light_para.parameters())},
             {'params': list(ns.light_inten_para.parameters()), 'lr': 1e-3}], lr=5e-4)
    else:  # trainer.py:154-163: no table, no parameter group; the run must never touch the attribute
        ns.light_optimizer = torch.optim.SparseAdam([{'params': list(ns.light_para.parameters())}], lr=5e-4)
    ns.light_scheduler = None
    class Loader(object):   # a DataLoader hands out fresh dictionaries every epoch (the loop edits them in place, trainer.py:365-367)
        def __len__(self):
            return len(batches)
        def __iter__(self):
            for idx, mi, gt in batches:
                yield idx, dict(mi), dict(gt)
    ns.train_dataloader = Loader()
    ns.train_dataset = types.SimpleNamespace(change_sampling_idx=lambda n: None, view_idx=[0, 1], light_slt=light_slt, visibility=view_vis,
                                             light_direction=[torch.nn.functional.normalize(li, dim=-1) for li in light_init])
    ns.num_pixels, ns.start_epoch, ns.nepochs = N, 1666, 1667      # cur_iter = 1666 * 3 = 4998 ... 5003: two epochs of three batches
    ns.ckpt_freq = ns.plot_freq = 10 ** 9
    ns.save_checkpoints, ns.plot_to_disk = (lambda e: None), (lambda: None)
    ns.obj_name, ns.expname = 'golden', 'golden'
    ns.writer = types.SimpleNamespace(add_scalar=lambda *a, **k: None)
    tmp = tempfile.mkdtemp()
    ns.checkpoints_path, ns.plots_dir = os.path.join(tmp, 'ckpt'), os.path.join(tmp, 'plots')
    os.makedirs(ns.checkpoints_path); os.makedirs(ns.plots_dir)
    ns.train_fix = lambda: RT.TrainRunner.train_fix(ns)
    # the state train_fix left at iteration 0 (trainer.py:486-504), produced by the reference's own method
    ns.cur_iter = 0
    RT.TrainRunner.train_fix(ns)
    assert ns.loss.vis_weight == 10 and ns.loss.sg_rgb_weight == 0 and not (light_train and ns.light_para.weight.requires_grad) \
        and not any(q.requires_grad for q in rnet.albedo_net.parameters()) and any(q.requires_grad for q in rnet.visibility_net.parameters()) == ns.vis_loss
    seed = 91
    torch.manual_seed(seed)
    np.random.seed(seed)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        RT.TrainRunner.run(ns)
    assert ns.cur_iter == 5004 and len(ns.loss.log) == 6
    assert ns.loss.sg_rgb_weight == 1.0 and all(q.requires_grad for q in rnet.albedo_net.parameters())
    assert (not light_train) or ns.light_para.weight.requires_grad == (not ns.ana_fixlight)
    assert any(q.requires_grad for q in rnet.visibility_net.parameters()) == ns.vis_loss   # (frozen for good without the loss, trainer.py:498-499)

    # the draws of the six forward passes (renderer.py:212: one torch.normal per step; normal jitter is off in bear.conf)
    torch.manual_seed(seed)
    noises = [torch.normal(0, torch.ones(int(batches[i % 3][1]['surface_mask'].sum()), 3) * 0.01) for i in range(6)]

    # oracle: the same six steps
    onet = o2.PSNetwork(conf)
    onet.load_state_dict(sd)
    o_vp = dict(light=vp_light, vis=vp_vis, view_light=light_init, view_vis=view_vis, vnum=vnum) if vis_plus else None
    ostep = o2.TrainStep(onet, conf, NL, torch.cat(light_init, dim=0), vis_plus=o_vp)
    np.random.seed(seed)
    ostep.cur_iter = 0
    ostep.train_fix()
    ostep.cur_iter = 4998
    accu = [len(l) for l in light_slt]
    olog, l_slts = [], []
    for i in range(6):
        idx, mi, gt = batches[i % 3]
        view = int(idx[0])
        l_slt = (sum(accu[:view]) + mi['lidx'][0]) if multi_light else torch.tensor([view])   # trainer.py:370-375
        l_slts.append(l_slt)
        inp_o = {k: v for k, v in mi.items() if k not in ('lidx', 'vidx', 'vidx_ori')}
        if multi_light:
            inp_o['light_direction'], inp_o['visibility'] = mi['light_direction'][0], mi['visibility'][0]
        t, _ = ostep.step(inp_o, {'rgb': gt['rgb'][0] if multi_light else gt['rgb']}, l_slt, train_order=True, noise={'xyz': noises[i]},
                          vidx=view if vis_plus else None)
        olog.append(t)
    keys = sorted(ns.loss.log[0])
    assert ns.loss.log[0]['albedo_smooth_loss'] is None and ns.loss.log[3]['albedo_smooth_loss'] is not None  # weight 0 before 5000
    for i in range(6):
        for k in keys:
            if ns.loss.log[i][k] is None:
                assert olog[i][k] is None, (i, k)
                continue
            check('trainer it %d %s' % (4998 + i, k), float(olog[i][k]), ns.loss.log[i][k], 2e-6)
        check('trainer it %d normal_loss' % (4998 + i), float(olog[i]['normal_loss']), ns.loss_n.log[i]['normal_loss'], 2e-6)
        assert ('vis_loss' in ns.loss.log[i]) == ns.vis_loss
    rsd, osd = rnet.state_dict(), onet.state_dict()
    # after Adam steps an element whose gradient sits at the fp32 noise floor may step the other way (|delta| <= 2 lr per step)
    for k in rsd:
        d = (rsd[k] - osd[k]).abs()
        assert float(d.max()) <= 2 * 6 * 5e-4 + 1e-6 and float(d.mean()) <= 2e-5, (k, float(d.max()), float(d.mean()))
    if light_train:
        check('trainer light table', ostep.light_para.weight.detach(), ns.light_para.weight.detach(), 1e-3)
        if ns.ana_fixlight:
            assert float((ns.light_para.weight.detach() - torch.cat(light_init, dim=0)).abs().max()) == 0.0
    else:
        assert not hasattr(ns, 'light_para') and float((ostep.light_para.weight.detach() - torch.cat(light_init, dim=0)).abs().max()) == 0.0
    if not light_train:
        pass
    elif inten_train:
        check('trainer light intensity', ostep.light_inten_para.weight.detach(), ns.light_inten_para.weight.detach(), 1e-3)
    else:
        assert not hasattr(ns, 'light_inten_para') and float((ostep.light_inten_para.weight.detach() - rnet.light_int).abs().max()) == 0.0
    moved = (sd['albedo_net.linears.0.weight'] - rsd['albedo_net.linears.0.weight']).abs().max()
    assert float(moved) > 0, 'the BRDF nets must have started training at iteration 5000'
    lk = sorted(keys)
    extra = {}
    if vis_plus:
        extra = dict(np_seed=seed, vnum=vnum, views=np.array([int(b[0][0]) for b in batches]), sampling_idx=np.stack([np_(x) for x in samp]),
                     vp_light=np.stack([np_(x) for x in vp_light]), vp_vis=np.stack([np_(x) for x in vp_vis]).astype(np.uint8),
                     view_vis0=np_(view_vis[0]).astype(np.uint8), view_vis1=np_(view_vis[1]).astype(np.uint8))
    np.savez_compressed(
        os.path.join(GOLDEN, trainer_fixture_name(vis_plus, inten_train, variant)), **extra,
        sd_digest=state_dict_digest(sd), N=N, L=L, V=V, NL=NL, input_seeds=np.array([200, 201, 202]), light_split=np.array([len(l) for l in light_slt]),
        **({} if variant is None else dict(variant=str(variant), overrides=json.dumps(over))),
        light_init=np_(torch.cat(light_init, dim=0)), l_slt=np.stack([np_(x) for x in l_slts]), first_iter=4998,
        noise0=np_(noises[0]), noise1=np_(noises[1]), noise2=np_(noises[2]), noise3=np_(noises[3]), noise4=np_(noises[4]), noise5=np_(noises[5]),
        # 'total' = what trainer.py:396-399 backpropagates (and, `loss += ...` being in place, what loss_output['loss'] holds afterwards)
        total=np.array([ns.loss.log[i]['loss'] + ns.loss_n.log[i]['loss'] for i in range(6)]),
        loss_names=np.array(lk + ['normal_loss']),
        loss_vals=np.array([[(np.nan if ns.loss.log[i][k] is None else ns.loss.log[i][k]) for k in lk] + [ns.loss_n.log[i]['normal_loss']]
                            for i in range(6)]),  # nan = the reference returned None (term switched off)
        light_para=np_(ns.light_para.weight.detach()) if light_train else np_(torch.cat(light_init, dim=0)),
        light_inten_para=np_(ns.light_inten_para.weight.detach()) if (inten_train and light_train) else np.full((NL, 1), rnet.light_int, dtype=np.float32),
        # final parameters: the first 2048 elements of every tensor (element-level check) + whole-tensor digests
        param_names=np.array(sorted(rsd)), param_norms=grad_digest(rsd)[1], param_projs=grad_digest(rsd)[2],
        **{('p_' + k): np_(v.reshape(-1)[:2048]) for k, v in rsd.items()})
    print('stage2 trainer golden written (vis_plus=%s, inten_train=%s, variant=%s)' % (vis_plus, inten_train, variant))


def gen_configs():
    """tests/golden/configs.json: the hot-path VALUES of all 14 reference configuration files (no file text).
    Stage 1: read by the reference's OWN loader (stage1/dataloading/configloading.py, imported as a file: it needs only
    PyYAML).  Stage 2: pyhocon is not in this image, so the .conf files are read by an independent line scanner below
    (section stack + ``key = value``; it shares no code with psnerf_amd.stage2.conf) -- the CPU tests then check
    psnerf_amd's readers, ``bear_conf()`` / ``object_conf()`` and ``stage1_cfg()`` against these values."""
    import importlib.util
    import json
    import re
    spec = importlib.util.spec_from_file_location('ref_configloading', os.path.join(REF, 'stage1/dataloading/configloading.py'))
    ref_cl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cl)
    from psnerf_amd.stage1.config import HOT_KEYS, hot_path, load_config
    out = {'stage1': {}, 'stage2': {}}
    for obj in ('armadillo', 'bear', 'buddha', 'bunny', 'cow', 'pot2', 'reading'):
        path = os.path.join(REF, 'stage1/configs/%s.yaml' % obj)
        cfg = ref_cl.load_config(path)
        out['stage1'][obj] = {sec: {k: cfg[sec][k] for k in keys if k in cfg.get(sec, {})} for sec, keys in HOT_KEYS.items()}
        assert hot_path(load_config(path)) == out['stage1'][obj], obj  # the build's reader == the reference's, here and now
        # independent scan of stage2/confs/<obj>.conf
        stack, flat = [], {}
        for raw in open(os.path.join(REF, 'stage2/confs/%s.conf' % obj)):
            line = raw.split('#')[0].strip()
            if not line:
                continue
            if line.endswith('{'):
                stack.append(line[:-1].strip())
            elif line == '}':
                stack.pop()
            else:
                k, v = [t.strip() for t in line.split('=', 1)]
                if re.fullmatch(r'\[.*\]', v):
                    val = [json.loads(t) for t in v[1:-1].split(',') if t.strip()]
                elif v in ('True', 'False', 'true', 'false'):
                    val = v.lower() == 'true'
                else:
                    try:
                        val = int(v)
                    except ValueError:
                        try:
                            val = float(v)
                        except ValueError:
                            val = v
                flat['.'.join(stack + [k])] = val
        assert not stack
        # everything but paths / names / plot cadence (which the hot path never reads)
        skip = ('dataset.', 'train.expname', 'train.stage1_shape_path', 'train.plot_freq', 'train.ckpt_freq', 'train.dataset_class')
        out['stage2'][obj] = {k: v for k, v in sorted(flat.items()) if not k.startswith(skip)}
    with open(os.path.join(GOLDEN, 'configs.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('configs golden written: %d stage-1 + %d stage-2 objects' % (len(out['stage1']), len(out['stage2'])))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    os.makedirs(GOLDEN, exist_ok=True)
    if what == 'all':  # separate processes: stage1 and stage2 both own a top-level ``utils``/``model`` package
        for s in ('stage1', 'stage2', 'trainer', 'configs'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), s])
    elif what == 'configs':
        gen_configs()
    elif what == 'trainer':
        gen_trainer(False)
        gen_trainer(True)
        gen_trainer(False, inten_train=False)
        for variant in ('gtlight', 'fixlight', 'novisloss'):
            gen_trainer(variant=variant)
    elif what == 'stage1':
        gen_stage1()
    elif what == 'stage2':
        gen_stage2()
