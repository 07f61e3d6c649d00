"""PSNR parity over a training HORIZON (BASELINE.json metric: "...; PSNR parity", north star: rendered PSNR within 0.05 dB).

Every other trajectory test follows HIP and oracle for 2 - 6 optimiser steps.  fp32 trajectories of two different summation
orders separate slowly over hundreds of steps; what must hold then is not elementwise agreement of the weights but (i) loss
curves that stay inside a band of each other and (ii) the same rendered quality at the end.  Both trainers start from the same
weights, see the same batches and the same injected random draws; the scenes are LEARNABLE (ground truth rendered by a teacher
network / a smooth image), so the PSNR really moves during the run and "equal PSNR" is not the trivial statement it would be on
noise targets.

    stage 2: 300 steps across the train_fix switch at iteration 5000 (stage2/trainer.py:355-410,485-513),
    stage 1: 200 steps with the normal loss on (stage1/model/training.py:46-60),
on scenes small enough for the CPU oracle (~1 min each on the GPU box's host).  PSN_CONVERGENCE_STEPS overrides the horizon."""
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import stage1_cfg, stage1_state_dict, stage2_state_dict
from psnerf_amd.synthetic import stage2_inputs

pytestmark = pytest.mark.gpu

PSNR_TOL_DB = 0.05  # north star
REPORT = bool(os.environ.get('PSN_PARITY_REPORT'))


def _psnr(a, b, mask=None):
    """stage2/trainer.py:268-276 (mse over the masked pixels, -10 log10)."""
    a, b = a.detach().double().cpu().reshape(-1, 3), b.detach().double().cpu().reshape(-1, 3)
    if mask is not None:
        m = mask.detach().cpu().reshape(-1).bool()
        a, b = a[m], b[m]
    mse = float(((a - b) ** 2).mean())
    return 100.0 if mse == 0 else -10.0 * math.log10(mse)


def test_stage2_300_steps_psnr_parity(cuda):
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    n_steps = int(os.environ.get('PSN_CONVERGENCE_STEPS', 300))
    N, L, V, n_views = 640, 8, 4, 3
    NL = L * n_views
    conf = o2.bear_conf()
    # ---- the scene: a teacher network renders the ground truth of three views under their own 8 lights ----------------
    teacher = o2.PSNetwork(conf)
    teacher.load_state_dict(stage2_state_dict(conf, seed=77))
    with torch.no_grad():  # (a freshly initialised visibility net answers ~0 everywhere: lift it so that the scene is lit)
        teacher.visibility_net.linears[-1].bias += 0.75
    views = []
    g = torch.Generator().manual_seed(5)
    for v in range(n_views):
        inp, gt = stage2_inputs(N, L, V, seed=300 + v)
        with torch.no_grad():
            t_out = teacher(inp, noise={'xyz': torch.zeros(int(inp['surface_mask'].sum()), 3)})
        gt = {'rgb': t_out['sg_rgb_values'].detach().clone()}
        sm = inp['surface_mask'][0]
        thr = t_out['vis_train'][:, sm, 0].median()  # binary supervision, as stage 1 hands it over: half of the pairs lit
        inp['vis_train_gt'] = (t_out['vis_train'][..., 0] > thr).float()
        inp['visibility'] = (t_out['visibility'][..., 0] > thr).float()
        true_dirs = inp.pop('light_direction')
        inp.pop('light_intensity')
        views.append((inp, gt, true_dirs))
    # SDPS-Net-like initial light estimates: the true directions, perturbed
    light_init = torch.nn.functional.normalize(torch.cat([t for _, _, t in views]) + 0.05 * torch.randn(NL, 3, generator=g), dim=-1)
    sd = stage2_state_dict(conf, seed=9)
    onet = o2.PSNetwork(conf)
    onet.load_state_dict(sd)
    ostep = o2.TrainStep(onet, conf, NL, light_init)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda)
    start = 5000 - n_steps // 2
    for tr in (ostep, step):  # the state train_fix left at iteration 0
        tr.cur_iter = start
        tr._ori = (1.0, 0.05, 0.01, 1)
        tr.loss.sg_rgb_weight, tr.loss.albedo_smooth_weight, tr.loss.rough_smooth_weight, tr.loss.vis_weight = 0, 0, 0, 10
        tr.model.albedo_net.eval().requires_grad_(False)
        tr.model.rough_net.eval().requires_grad_(False)
        tr.light_para.requires_grad_(False)
        tr.light_inten_para.requires_grad_(False)
    views_d = [({k: t.to(cuda) for k, t in inp.items()}, {k: t.to(cuda) for k, t in gt.items()}) for inp, gt, _ in views]

    def render_psnr(model, tr, dev):
        vals = []
        with torch.no_grad():
            for v, (inp, gt, _) in enumerate(views):
                mi = {k: t.to(dev) for k, t in inp.items()}
                l_slt = torch.arange(L, device=dev) + L * v
                mi['light_direction'] = torch.nn.functional.normalize(tr.light_para.weight.detach()[l_slt], dim=-1)
                mi['light_intensity'] = tr.light_inten_para.weight.detach()[l_slt]
                out = model(mi, noise={'xyz': torch.zeros(int(inp['surface_mask'].sum()), 3, device=dev)})
                m = (inp['surface_mask'] & inp['object_mask']).expand(L, -1)
                vals.append(_psnr(out['sg_rgb_values'], gt['rgb'], m))
        return float(np.mean(vals))

    psnr0 = render_psnr(onet, ostep, 'cpu')
    lo, lh = [], []
    for it in range(n_steps):
        v = it % n_views
        inp, gt, _ = views[v]
        l_slt = torch.arange(L) + L * v
        nz = torch.randn(int(inp['surface_mask'].sum()), 3, generator=g) * 0.01
        ot, _ = ostep.step(inp, gt, l_slt, noise={'xyz': nz})
        pt, _ = step.step(views_d[v][0], views_d[v][1], l_slt.to(cuda), noise={'xyz': nz.to(cuda)})
        lo.append(float(ot['total']))
        lh.append(float(pt['total'].detach()))
    assert step.cur_iter == ostep.cur_iter == start + n_steps and step.cur_iter > 5000
    lo, lh = np.array(lo), np.array(lh)
    rel = np.abs(lh - lo) / np.abs(lo)
    psnr_o, psnr_h = render_psnr(onet, ostep, 'cpu'), render_psnr(net, step, cuda)
    if REPORT:
        print('stage2 convergence: PSNR %.3f -> oracle %.4f / HIP %.4f dB (diff %.4f); loss %.4f -> %.4f; rel loss diff max %.2e, '
              'last-50 mean %.2e' % (psnr0, psnr_o, psnr_h, psnr_h - psnr_o, lo[0], lo[-1], rel.max(), rel[-50:].mean()))
    assert np.isfinite(lh).all()
    # the scene is learnable: the run must have moved the PSNR (otherwise equal PSNR would say nothing)
    assert psnr_o > psnr0 + 1.0, (psnr0, psnr_o)
    # loss curves: every step within 2 % of the oracle's, the phase-2 tail within 0.5 % on average (first steps: 1e-4)
    assert rel[:3].max() <= 2e-4, rel[:3]
    assert rel.max() <= 2e-2, rel.max()
    assert rel[-50:].mean() <= 5e-3, rel[-50:].mean()
    assert abs(psnr_h - psnr_o) <= PSNR_TOL_DB, (psnr_h, psnr_o)


def test_stage1_200_steps_psnr_parity(cuda):
    from oracle import stage1 as o1
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.stage1.rendering import sync_free_noise_for_reference
    from psnerf_amd.synthetic import stage1_batch
    n_steps = int(os.environ.get('PSN_CONVERGENCE_STEPS', 200))
    R, h, w = 64, 32, 40
    cfg = stage1_cfg('bunny', **{'training.n_training_points': R})
    sd = stage1_state_dict(cfg, seed=21)
    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(sd)
    oren = o1.Renderer(onet, cfg)
    otr = o1.Trainer(oren, torch.optim.Adam(onet.parameters(), lr=1e-4), cfg)
    net = NeuralNetwork(cfg)
    net.load_state_dict(sd)
    ren = Renderer(net, cfg, device=cuda)
    tr = Trainer(ren, torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=cuda)
    # ---- a learnable target: smooth colours, the silhouette of the initial shape as the mask -----------------------------
    batch = stage1_batch(cfg, h=h, w=w, seed=4)
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    batch['img'] = torch.stack([0.5 + 0.4 * torch.sin(xx / 9.0), 0.5 + 0.4 * torch.cos(yy / 7.0), 0.3 + 0.3 * torch.sin((xx + yy) / 11.0)])[None]
    grid = torch.stack([xx.reshape(-1), yy.reshape(-1)], -1)[None]  # [1, h*w, 2] (x, y)
    cam = (batch['img.camera_mat'], batch['img.world_mat'], batch['img.scale_mat'])
    with torch.no_grad():
        sil = oren(grid, *cam, 'unisurf', add_noise=False, eval_=True, it=1000)['mask_pred']
    batch['img.mask'] = sil.reshape(1, h, w).float()
    batch_d = {k: v.to(cuda) for k, v in batch.items()}
    gt_rgb = batch['img'][0].permute(1, 2, 0).reshape(-1, 3)

    def render_psnr(renderer, dev):
        with torch.no_grad():
            out = renderer(grid.to(dev), *[c.to(dev) for c in cam], 'unisurf', add_noise=False, eval_=True, it=2000)
        return _psnr(out['rgb'], gt_rgb)

    psnr0 = render_psnr(oren, 'cpu')
    gen = torch.Generator().manual_seed(8)
    lo, lh, flips = [], [], 0
    for k in range(n_steps):
        it = 1000 + k
        pix = torch.stack([torch.randint(0, w, (R,), generator=gen).float(), torch.randint(0, h, (R,), generator=gen).float()], -1)[None]
        with torch.no_grad():  # the hit mask of the oracle's CURRENT weights splits the per-ray tables into the reference's draw order
            dry = oren(pix, *cam, 'unisurf', add_noise=False, eval_=True, it=it)
        noise = {'full': torch.rand(R, 64, generator=gen), 'nbr_full': torch.rand(R, 3, generator=gen)}
        ot = otr.train_step(batch, it=it, pix=pix, noise=sync_free_noise_for_reference(noise, dry['mask_pred']))
        pt = tr.train_step(batch_d, it=it, pix=pix.to(cuda), noise={n: t.to(cuda) for n, t in noise.items()})
        lo.append(float(ot['loss'].detach()))
        lh.append(float(pt['loss'].detach()))
    lo, lh = np.array(lo), np.array(lh)
    rel = np.abs(lh - lo) / np.abs(lo)
    psnr_o, psnr_h = render_psnr(oren, 'cpu'), render_psnr(ren, cuda)
    if REPORT:
        print('stage1 convergence: PSNR %.3f -> oracle %.4f / HIP %.4f dB (diff %.4f); loss %.4f -> %.4f; rel loss diff max %.2e, '
              'last-50 mean %.2e' % (psnr0, psnr_o, psnr_h, psnr_h - psnr_o, lo[0], lo[-1], rel.max(), rel[-50:].mean()))
    assert np.isfinite(lh).all()
    assert psnr_o > psnr0 + 1.0, (psnr0, psnr_o)
    assert rel[:2].max() <= 5e-4, rel[:2]
    # a ray whose surface crossing flips between hit and miss under 1e-6 weight differences changes its sample set, so
    # single steps may differ visibly; the band is on every step and tighter on the tail average
    assert rel.max() <= 5e-2, rel.max()
    assert rel[-50:].mean() <= 1e-2, rel[-50:].mean()
    assert abs(psnr_h - psnr_o) <= PSNR_TOL_DB, (psnr_h, psnr_o)
