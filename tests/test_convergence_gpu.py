"""PSNR parity over a training HORIZON (BASELINE.json metric: "...; PSNR parity", north star: rendered PSNR within 0.05 dB).

Every other trajectory test follows HIP and oracle for 2 - 6 optimiser steps; here: 300 stage-2 steps across the train_fix
switch at iteration 5000 (stage2/trainer.py:355-410,485-513) and 200 stage-1 steps with the normal loss on
(stage1/model/training.py:46-60), on scenes small enough for the CPU oracle.  Same initial weights, same batches, same
injected random draws on both sides.

What can and what cannot hold over such a horizon was MEASURED first (tools/dbg/convergence_control.py): the
REFERENCE arithmetic run against ITSELF with its initial weights perturbed by 1e-7 relative (one fp32 ulp) follows the same loss
curve to < 1e-3 for ~40 steps and then separates exponentially -- single-step losses differ by up to 50 % around step 100, the
tail averages agree to ~1 %, the final PSNR of five such replicas spreads over 0.21 dB (sigma 0.075 dB) on the stage-2 scene
below.  Adam on ReLU / softplus networks amplifies rounding differences; "PSNR within 0.05 dB after free-running training" is
therefore not a property the reference has with respect to itself, and the tests are built accordingly:

  * SYNCHRONISED windows (the deterministic statement): every 10 steps the HIP trainer takes over the oracle's complete state
    (weights, light tables, Adam / SparseAdam moments and step counts) and both continue from it -- 30 / 20 different states
    along the whole horizon, both train_fix phases.  ONE step away from a common state the losses agree to 1e-5 (measured
    2e-7) and the two models RENDER THE SAME PSNR within 0.05 dB (measured 0.002 dB); ten free steps later the stage-2 models
    still render within 0.05 dB (measured 0.015 dB), the stage-1 models -- whose PSNR climbs 0.09 dB per step -- within 0.5 dB
    (measured 0.11 dB).
  * FREE-RUNNING (the statistical statement, stage 2): HIP runs the 300 steps on its own; its final PSNR and tail loss must lie
    within the spread of the reference arithmetic against itself, measured in the same test by two 1e-7-perturbed oracle
    replicas: |PSNR(HIP) - PSNR(oracle)| <= 0.05 dB + 2 x (range of the three oracle runs).
PSN_CONVERGENCE_STEPS overrides the horizon, PSN_PARITY_REPORT=1 prints the measured numbers."""
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
DUMP = os.environ.get('PSN_CONVERGENCE_DUMP')  # directory: the loss curves of both trainers as .npz (calibration of the bands)


@pytest.fixture(autouse=True)
def _oracle_threads():
    """The CPU oracle is eager torch: on the 256-core host of the GPU box it is fastest with 16 threads (bench.py's sweep)."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(16, n))
    yield
    torch.set_num_threads(n)


def _dump(name, **arrays):
    if DUMP:
        os.makedirs(DUMP, exist_ok=True)
        np.savez(os.path.join(DUMP, name + '.npz'), **arrays)


def _psnr(a, b, mask=None):
    """stage2/trainer.py:268-276 (mse over the masked pixels, -10 log10)."""
    a, b = a.detach().double().cpu().reshape(-1, 3), b.detach().double().cpu().reshape(-1, 3)
    if mask is not None:
        m = mask.detach().cpu().reshape(-1).bool()
        a, b = a[m], b[m]
    mse = float(((a - b) ** 2).mean())
    return 100.0 if mse == 0 else -10.0 * math.log10(mse)


def _stage2_scene():
    """Three views rendered by a teacher network under their own 8 lights; SDPS-Net-like perturbed initial light estimates."""
    from oracle import stage2 as o2
    N, L, V, n_views = 640, 8, 4, 3
    conf = o2.bear_conf()
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
    NL = L * n_views
    light_init = torch.nn.functional.normalize(torch.cat([t for _, _, t in views]) + 0.05 * torch.randn(NL, 3, generator=g), dim=-1)
    return views, light_init, L, NL


def _phase1(tr, start):
    """The state train_fix left at iteration 0 (trainer.py:485-499), at iteration ``start``."""
    tr.cur_iter = start
    tr._ori = (1.0, 0.05, 0.01, 1)
    tr.loss.sg_rgb_weight, tr.loss.albedo_smooth_weight, tr.loss.rough_smooth_weight, tr.loss.vis_weight = 0, 0, 0, 10
    tr.model.albedo_net.eval().requires_grad_(False)
    tr.model.rough_net.eval().requires_grad_(False)
    tr.light_para.requires_grad_(False)
    tr.light_inten_para.requires_grad_(False)


def _stage2_render_psnr(model, tr, views, L, dev):
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


def _oracle_trainer(sd, light_init, NL, start):
    from oracle import stage2 as o2
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    ostep = o2.TrainStep(onet, o2.bear_conf(), NL, light_init)
    _phase1(ostep, start)
    return onet, ostep


def _stage2_draws(views, n_steps, seed=6):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(int(views[it % len(views)][0]['surface_mask'].sum()), 3, generator=g) * 0.01 for it in range(n_steps)]


@pytest.fixture(params=['fp32', 'bf16x6', 'chains_bf16x3'])   # (the three-product weight gradients alone, 'bf16x3', are a subset of 'chains_bf16x3': dropped in round 6, -77 s)
def wgrad(request):
    """The synchronised-window tests run with the exact fp32 weight-gradient kernel, with the split-bf16 experiments
    (psn_gemm_tn_grouped_x3 through hip.wgrad_precision: six / three partial products) and with the chains' matrix work on the
    bf16 pipe as well (ops.chain_precision('bf16x3') beside the three-product weight gradients) -- the same bounds hold for all."""
    from psnerf_amd import hip, ops
    if request.param == 'chains_bf16x3':
        with hip.wgrad_precision('bf16x3'), ops.chain_precision('bf16x3'):
            yield request.param
    else:
        with hip.wgrad_precision(request.param):
            yield request.param


def test_stage2_300_steps_synchronised_windows(cuda, wgrad):
    import psnerf_amd.stage2 as s2
    n_steps, W = int(os.environ.get('PSN_CONVERGENCE_STEPS', 300)), 10
    views, light_init, L, NL = _stage2_scene()
    sd = stage2_state_dict(__import__('oracle.stage2', fromlist=['x']).bear_conf(), seed=9)
    start = 5000 - n_steps // 2
    onet, ostep = _oracle_trainer(sd, light_init, NL, start)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda)
    _phase1(step, start)
    views_d = [({k: t.to(cuda) for k, t in inp.items()}, {k: t.to(cuda) for k, t in gt.items()}) for inp, gt, _ in views]
    draws = _stage2_draws(views, n_steps)
    psnr0 = _stage2_render_psnr(onet, ostep, views, L, 'cpu')
    rel, dps, dps1, pdiff = [], [], [], []
    for it in range(n_steps):
        v = it % len(views)
        l_slt = torch.arange(L) + L * v
        ot, _ = ostep.step(views[v][0], views[v][1], l_slt, noise={'xyz': draws[it]})
        pt, _ = step.step(views_d[v][0], views_d[v][1], l_slt.to(cuda), noise={'xyz': draws[it].to(cuda)})
        rel.append(abs(float(pt['total'].detach()) - float(ot['total'].detach())) / abs(float(ot['total'].detach())))
        if it % W == 0:  # one optimiser step away from a common state: the two models render the same image
            dps1.append(_stage2_render_psnr(net, step, views, L, cuda) - _stage2_render_psnr(onet, ostep, views, L, 'cpu'))
            osd = onet.state_dict()
            pdiff.append(max(float((t.cpu() - osd[k]).abs().max()) for k, t in net.state_dict().items()))
        if (it + 1) % W == 0 or it == n_steps - 1:
            po, ph = _stage2_render_psnr(onet, ostep, views, L, 'cpu'), _stage2_render_psnr(net, step, views, L, cuda)
            dps.append(ph - po)
            osd = onet.state_dict()
            # the HIP trainer takes over the oracle's complete state
            net.load_state_dict(osd)
            step.light_para.weight.data.copy_(ostep.light_para.weight.detach())
            step.light_inten_para.weight.data.copy_(ostep.light_inten_para.weight.detach())
            step.sg_optimizer.load_state_dict(ostep.sg_optimizer.state_dict())
            step.light_optimizer.load_state_dict(ostep.light_optimizer.state_dict())
            net.invalidate_packs()
    assert step.cur_iter == ostep.cur_iter == start + n_steps and step.cur_iter > 5000
    rel, dps, dps1 = np.array(rel), np.array(dps), np.array(dps1)
    psnr1 = _stage2_render_psnr(onet, ostep, views, L, 'cpu')
    _dump('stage2_windows', rel=rel, dps=dps, dps1=dps1, pdiff=np.array(pdiff))
    if REPORT:
        print('stage2 synchronised windows: PSNR %.3f -> %.3f dB; rel loss diff: first step of a window max %.2e, all steps max %.2e '
              '(median %.2e); PSNR diff one step after a sync: max |d| %.5f dB, at the %d window ends: max |d| %.4f dB; max |param diff| '
              'one step after a sync %.2e' % (psnr0, psnr1, rel[::W].max(), rel.max(), np.median(rel), np.abs(dps1).max(), len(dps),
                                              np.abs(dps).max(), max(pdiff)))
    assert np.isfinite(rel).all() and abs(psnr1 - psnr0) > 0.2  # (the run moved the rendering)
    # from each of the 30 states along the horizon, ONE step: same loss, same rendering, parameters within Adam's sign-flip bound
    assert rel[::W].max() <= 1e-5, rel[::W]
    assert np.abs(dps1).max() <= PSNR_TOL_DB, dps1
    assert max(pdiff) <= 3 * 5e-4, pdiff  # an element whose gradient is at the fp32 noise floor may step the other way: 2 lr (measured 1.0e-3)
    # inside the 10-step windows the two trajectories separate at the reference arithmetic's own rate (module docstring);
    # measured: max 9.6e-3, median 9e-5, PSNR at the 30 window ends within 0.015 dB
    assert np.median(rel) <= 1e-3 and rel.max() <= 5e-2, (np.median(rel), rel.max())
    assert np.abs(dps).max() <= PSNR_TOL_DB, dps


def test_stage2_300_steps_free_running_within_reference_spread(cuda):
    import psnerf_amd.stage2 as s2
    n_steps = int(os.environ.get('PSN_CONVERGENCE_STEPS', 300))
    views, light_init, L, NL = _stage2_scene()
    sd = stage2_state_dict(__import__('oracle.stage2', fromlist=['x']).bear_conf(), seed=9)
    start = 5000 - n_steps // 2
    draws = _stage2_draws(views, n_steps)

    def run_oracle(state):
        onet, ostep = _oracle_trainer(state, light_init, NL, start)
        losses = []
        for it in range(n_steps):
            v = it % len(views)
            ot, _ = ostep.step(views[v][0], views[v][1], torch.arange(L) + L * v, noise={'xyz': draws[it]})
            losses.append(float(ot['total'].detach()))
        return np.array(losses), _stage2_render_psnr(onet, ostep, views, L, 'cpu')

    lo, psnr_o = run_oracle(sd)
    # the reference arithmetic against itself: initial weights perturbed by one fp32 ulp (1e-7 relative)
    reps = []
    for seed in (1, 2):
        gp = torch.Generator().manual_seed(seed)
        sd_p = {k: (t * (1 + 1e-7 * torch.randn(t.shape, generator=gp)) if t.dtype.is_floating_point else t) for k, t in sd.items()}
        reps.append(run_oracle(sd_p))
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda)
    _phase1(step, start)
    views_d = [({k: t.to(cuda) for k, t in inp.items()}, {k: t.to(cuda) for k, t in gt.items()}) for inp, gt, _ in views]
    lh = []
    for it in range(n_steps):
        v = it % len(views)
        pt, _ = step.step(views_d[v][0], views_d[v][1], (torch.arange(L) + L * v).to(cuda), noise={'xyz': draws[it].to(cuda)})
        lh.append(float(pt['total'].detach()))
    lh = np.array(lh)
    psnr_h = _stage2_render_psnr(net, step, views, L, cuda)
    ps = np.array([psnr_o] + [r[1] for r in reps])
    tails = np.array([lo[-50:].mean()] + [r[0][-50:].mean() for r in reps])
    spread, tail_spread = float(ps.max() - ps.min()), float(tails.max() - tails.min())
    rel = np.abs(lh - lo) / np.abs(lo)
    rel_rep = max(float((np.abs(r[0] - lo) / np.abs(lo))[:30].max()) for r in reps)
    _dump('stage2_free', lo=lo, lh=lh, reps=np.stack([r[0] for r in reps]), psnr=np.concatenate([ps, [psnr_h]]))
    if REPORT:
        print('stage2 free-running: PSNR oracle %.4f, replicas %s, HIP %.4f dB (spread of the reference against itself %.4f dB); tail '
              'loss oracle %.5f replicas %s HIP %.5f; first-30-step rel diff HIP %.2e, replicas %.2e'
              % (psnr_o, np.round(ps[1:], 4), psnr_h, spread, tails[0], np.round(tails[1:], 5), lh[-50:].mean(), rel[:30].max(), rel_rep))
    assert np.isfinite(lh).all()
    assert rel[:30].max() <= 1e-3, rel[:30].max()  # before the trajectories separate: the same curve
    assert abs(psnr_h - psnr_o) <= PSNR_TOL_DB + 2.0 * spread, (psnr_h, ps)
    assert abs(lh[-50:].mean() - tails[0]) <= 0.01 * tails[0] + 2.0 * tail_spread, (lh[-50:].mean(), tails)


def test_stage1_200_steps_synchronised_windows(cuda, wgrad):
    from oracle import stage1 as o1
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.stage1.rendering import sync_free_noise_for_reference
    from psnerf_amd.synthetic import stage1_batch
    n_steps, W = int(os.environ.get('PSN_CONVERGENCE_STEPS', 200)), 10
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
    # the PSNR is rendered on every second pixel of every second row (320 rays: the oracle renders it 40 times)
    sub = (torch.arange(0, h, 2)[:, None] * w + torch.arange(0, w, 2)[None, :]).reshape(-1)
    grid_s = grid[:, sub]
    gt_rgb = batch['img'][0].permute(1, 2, 0).reshape(-1, 3)[sub]

    def render_psnr(renderer, dev):
        with torch.no_grad():
            out = renderer(grid_s.to(dev), *[c.to(dev) for c in cam], 'unisurf', add_noise=False, eval_=True, it=2000)
        return _psnr(out['rgb'], gt_rgb)

    psnr0 = render_psnr(oren, 'cpu')
    gen = torch.Generator().manual_seed(8)
    rel, dps, dps1 = [], [], []
    for k in range(n_steps):
        it = 1000 + k
        pix = torch.stack([torch.randint(0, w, (R,), generator=gen).float(), torch.randint(0, h, (R,), generator=gen).float()], -1)[None]
        with torch.no_grad():  # the hit mask of the oracle's CURRENT weights splits the per-ray tables into the reference's draw order
            dry = oren(pix, *cam, 'unisurf', add_noise=False, eval_=True, it=it)
        noise = {'full': torch.rand(R, 64, generator=gen), 'nbr_full': torch.rand(R, 3, generator=gen)}
        ot = otr.train_step(batch, it=it, pix=pix, noise=sync_free_noise_for_reference(noise, dry['mask_pred']))
        pt = tr.train_step(batch_d, it=it, pix=pix.to(cuda), noise={n: t.to(cuda) for n, t in noise.items()})
        rel.append(abs(float(pt['loss'].detach()) - float(ot['loss'].detach())) / abs(float(ot['loss'].detach())))
        if k % W == 0:  # one optimiser step away from a common state
            dps1.append(render_psnr(ren, cuda) - render_psnr(oren, 'cpu'))
        if (k + 1) % W == 0 or k == n_steps - 1:
            dps.append(render_psnr(ren, cuda) - render_psnr(oren, 'cpu'))
            net.load_state_dict(onet.state_dict())  # the HIP trainer takes over the oracle's complete state
            tr.optimizer.load_state_dict(otr.optimizer.state_dict())
            net.invalidate_packs()
    rel, dps, dps1 = np.array(rel), np.array(dps), np.array(dps1)
    psnr1 = render_psnr(oren, 'cpu')
    _dump('stage1_windows', rel=rel, dps=dps, dps1=dps1)
    if REPORT:
        print('stage1 synchronised windows: PSNR %.3f -> %.3f dB; rel loss diff: first step of a window max %.2e, all steps max %.2e '
              '(median %.2e); PSNR diff one step after a sync: max |d| %.5f dB, at the %d window ends: max |d| %.4f dB'
              % (psnr0, psnr1, rel[::W].max(), rel.max(), np.median(rel), np.abs(dps1).max(), len(dps), np.abs(dps).max()))
    assert np.isfinite(rel).all() and psnr1 > psnr0 + 5.0, (psnr0, psnr1)  # the target is learnable: the PSNR really moved
    assert rel[::W].max() <= 1e-5, rel[::W]
    assert np.abs(dps1).max() <= PSNR_TOL_DB, dps1
    # inside the windows (measured: max 1.4e-2, median 1.4e-4; the PSNR climbs 0.09 dB PER STEP on this scene -- 7.4 -> 24.8 dB
    # in 200 steps --, so ten free steps leave up to 0.11 dB between the two models at a window end)
    assert np.median(rel) <= 2e-3 and rel.max() <= 0.1, (np.median(rel), rel.max())
    assert np.abs(dps).max() <= 0.5, dps
