"""Stage-2 product path (HIP, through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle.  Tolerances: 1e-4 relative on outputs / losses (north star),
1e-3 on parameter-gradient digests (accumulation order)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, assert_close, assert_outputs_close, grad_digest, stage2_state_dict, stage2_truth, state_dict_digest
from psnerf_amd.synthetic import stage2_inputs

pytestmark = pytest.mark.gpu


def _run(net, MainLoss, NormalLoss, inp, gt, phase, noise, dev):
    if phase == 1:
        lw = dict(sg_rgb_weight=0, albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=10)
        net.albedo_net.eval().requires_grad_(False)
        net.rough_net.eval().requires_grad_(False)
    else:
        lw = dict(sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)
    inp = {k: v.to(dev) for k, v in inp.items()}
    gt = {k: v.to(dev) for k, v in gt.items()}
    ldir = inp['light_direction'].clone().requires_grad_(phase == 2)
    lint = inp['light_intensity'].clone().requires_grad_(phase == 2)
    inp['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
    inp['light_intensity'] = lint
    out = net(inp, noise={'xyz': noise.to(dev)})
    t = dict(MainLoss(loss_type='L1', **lw)(out, gt, inp))
    tn = NormalLoss(1, 0.05)(out)
    t['normal_loss'] = tn['normal_loss']
    t['total'] = t['loss'] + tn['loss']
    t['total'].backward()
    gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    if phase == 2:
        gr['__light_dir'], gr['__light_int'] = ldir.grad, lint.grad
    return out, t, gr


def _truth(conf, sd, inp, nz):
    """float64 oracle evaluation of what _run feeds the model (light directions normalised in fp32 first)."""
    from oracle import stage2 as o2
    onet = o2.PSNetwork(conf)
    onet.load_state_dict(sd)
    inp = dict(inp)
    inp['light_direction'] = torch.nn.functional.normalize(inp['light_direction'], p=2, dim=-1)
    return stage2_truth(onet, inp, noise={'xyz': nz})


@pytest.mark.parametrize('L', [1, 10])
@pytest.mark.parametrize('phase', [1, 2])
def test_psnetwork_golden(cuda, L, phase):
    import psnerf_amd.stage2 as s2
    g = np.load(os.path.join(GOLDEN, 'stage2_psnet_L%d_ph%d.npz' % (L, phase)))
    conf = s2.bear_conf()
    from oracle import stage2 as o2
    sd = stage2_state_dict(o2.bear_conf(), seed=31)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = s2.PSNetwork(conf)
    net.load_state_dict(sd)
    net.to(cuda)
    inp, gt = stage2_inputs(int(g['N']), L, int(g['V']), seed=int(g['input_seed']))
    out, t, gr = _run(net, s2.MainLoss, s2.NormalLoss, inp, gt, phase, torch.from_numpy(g['nz_xyz']), cuda)
    truth = _truth(o2.bear_conf(), sd, inp, torch.from_numpy(g['nz_xyz']))
    for k in g.files:
        if k.startswith('out_'):
            assert_outputs_close(k[4:], out[k[4:]].detach().cpu(), g[k], truth=truth)
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)].detach()), v, 1e-4, str(k), atol=0.0)
    names, norms, projs = grad_digest(gr)
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 1e-3, 'grad norms')
    assert_close(projs, g['grad_projs'], 1e-3, 'grad projs')
    if phase == 2:
        assert_close(gr['__light_dir'].cpu(), g['g_light_dir'], 1e-3, 'light dir grad')
        assert_close(gr['__light_int'].cpu(), g['g_light_int'], 1e-3, 'light int grad')


@pytest.mark.parametrize('N,L,V', [(3000, 5, 8), (777, 96, 3), (64, 2, 1)])
def test_psnetwork_vs_oracle(cuda, N, L, V):
    """Ragged sizes (not multiples of the 128-row tiles), per-parameter gradients against the oracle."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    inp, gt = stage2_inputs(N, L, V, seed=N)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    o_out, o_t, o_g = _run(onet, o2.MainLoss, o2.NormalLoss, inp, gt, 2, nz, 'cpu')
    out, t, gr = _run(net, s2.MainLoss, s2.NormalLoss, inp, gt, 2, nz, cuda)
    truth = _truth(o2.bear_conf(), sd, inp, nz)
    for k in o_out:
        if torch.is_tensor(o_out[k]) and o_out[k].dtype.is_floating_point:
            assert_outputs_close(k, out[k].detach().cpu(), o_out[k].detach(), truth=truth)
    for k in o_t:
        if o_t[k] is not None:
            assert_close(float(t[k]), float(o_t[k]), 1e-4, k, atol=0.0)
    assert sorted(gr.keys()) == sorted(o_g.keys())
    for k in o_g:
        assert_close(gr[k].cpu(), o_g[k], 1e-3, 'grad ' + k)
    # PSNR parity: the rendering metric of the two implementations against the same ground truth
    m = inp['surface_mask'][0]
    p_hip = s2.psnr(out['sg_rgb_values'].detach().cpu()[:, m], gt['rgb'][:, m])
    p_ref = o2.psnr(o_out['sg_rgb_values'].detach()[:, m], gt['rgb'][:, m])
    assert abs(p_hip - p_ref) < 1e-3, 'PSNR %.5f dB vs %.5f dB' % (p_hip, p_ref)


def test_empty_surface(cuda):
    import psnerf_amd.stage2 as s2
    net = s2.PSNetwork(s2.bear_conf()).to(cuda)
    inp, gt = stage2_inputs(100, 3, 2, seed=1, surface_frac=0.0, device=cuda)
    inp['surface_mask'][:] = False
    out = net(inp)
    assert float((out['sg_rgb_values'] - 1).abs().max()) == 0  # dense outputs are pre-filled with ones
    t = s2.MainLoss(1.0, 'L1', 0.05, 0.01, 1)(out, gt, inp)
    assert float(t['loss']) == 0


def test_train_steps_match_oracle(cuda):
    """Three optimisation steps across the train_fix switch (iteration 5000): parameters after the
    steps match the CPU oracle's within fp32 tolerance."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    sd = stage2_state_dict(o2.bear_conf(), seed=9)
    N, L, V, NL = 600, 6, 4, 40
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    ostep = o2.TrainStep(onet, o2.bear_conf(), NL, light_init)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda)
    ostep.cur_iter = step.cur_iter = 4999
    for tr in (ostep, step):  # emulate the state train_fix left at iteration 0
        tr._ori = (1.0, 0.05, 0.01, 1)
        tr.loss.sg_rgb_weight, tr.loss.albedo_smooth_weight, tr.loss.rough_smooth_weight, tr.loss.vis_weight = 0, 0, 0, 10
        tr.model.albedo_net.eval().requires_grad_(False)
        tr.model.rough_net.eval().requires_grad_(False)
        tr.light_para.requires_grad_(False)
        tr.light_inten_para.requires_grad_(False)
    for it in range(3):
        inp, gt = stage2_inputs(N, L, V, seed=100 + it)
        l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L]
        ns = int(inp['surface_mask'].sum())
        nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
        ot, _ = ostep.step(inp, gt, l_slt, noise={'xyz': nz})
        inp_d = {k: v.to(cuda) for k, v in inp.items()}
        gt_d = {k: v.to(cuda) for k, v in gt.items()}
        pt, _ = step.step(inp_d, gt_d, l_slt.to(cuda), noise={'xyz': nz.to(cuda)})
        assert_close(float(pt['total']), float(ot['total']), 2e-4, 'loss it%d' % it, atol=0.0)
    # Adam's first steps move every weight by ~lr*sign(g): elements whose gradient is at the fp32 noise
    # floor may legitimately step the other way (|delta| <= 2*lr per step), so bound the max by that and
    # require the bulk (mean abs difference) to agree tightly.
    osd = onet.state_dict()
    for k, v in net.state_dict().items():
        d = (v.cpu() - osd[k]).abs()
        assert float(d.max()) <= 3 * 2 * 5e-4 + 1e-6, 'param %s max diff %.3e' % (k, float(d.max()))
        assert float(d.mean()) <= 2e-5, 'param %s mean diff %.3e' % (k, float(d.mean()))
    assert_close(step.light_para.weight.detach().cpu(), ostep.light_para.weight.detach(), 1e-4, 'light dirs')
    assert_close(step.light_inten_para.weight.detach().cpu(), ostep.light_inten_para.weight.detach(), 1e-4, 'light int')


def test_side_stream_overlap_is_bit_identical(cuda):
    """The BRDF / normal networks on a side stream beside the visibility launch (train.overlap_small_nets, default on)
    change WHEN kernels run, never what they compute: six optimisation steps (both train_fix phases, 20k pixels so that
    the launches really overlap) give bit-identical losses, parameters, light tables and Adam states with the overlap
    switched off."""
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    sd = stage2_state_dict(conf, seed=9)
    N, L, V, NL = 20000, 12, 4, 40
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    res = {}
    for overlap in (True, False):
        net = s2.PSNetwork(conf)
        net.load_state_dict(sd)
        net.to(cuda)
        assert net.overlap_small_nets is True  # default
        net.overlap_small_nets = overlap
        step = s2.TrainStep(net, conf, NL, light_init.to(cuda), cuda)
        step.cur_iter = 4997
        step._ori = (1.0, 0.05, 0.01, 1)  # the state train_fix left at iteration 0 (as in test_train_steps_match_oracle)
        step.loss.sg_rgb_weight, step.loss.albedo_smooth_weight, step.loss.rough_smooth_weight, step.loss.vis_weight = 0, 0, 0, 10
        step.model.albedo_net.eval().requires_grad_(False)
        step.model.rough_net.eval().requires_grad_(False)
        step.light_para.requires_grad_(False)
        step.light_inten_para.requires_grad_(False)
        losses = []
        for it in range(6):
            inp, gt = stage2_inputs(N, L, V, seed=100 + it)
            l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L]
            nz = torch.randn(int(inp['surface_mask'].sum()), 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
            pt, _ = step.step({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()},
                              l_slt.to(cuda), noise={'xyz': nz.to(cuda)})
            losses.append(pt['total'].detach().clone())
        torch.cuda.synchronize()
        res[overlap] = (torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                        step.light_para.weight.detach().cpu().clone(), step.light_inten_para.weight.detach().cpu().clone())
    (l1, p1, a1, b1), (l0, p0, a0, b0) = res[True], res[False]
    assert torch.equal(l1, l0), (l1, l0)
    for k in p1:
        assert torch.equal(p1[k], p0[k]), k
    assert torch.equal(a1, a0) and torch.equal(b1, b0)


def test_psnetwork_microfacet_golden(cuda):
    """train.render_model = microfacet (GGX, stage2/model/microfacet.py) through the fused mf_shade kernel."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    g = np.load(os.path.join(GOLDEN, 'stage2_psnet_microfacet.npz'))
    sd = stage2_state_dict(o2.bear_conf(**{'train.render_model': 'microfacet'}), seed=33)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = s2.PSNetwork(s2.bear_conf(**{'train.render_model': 'microfacet'}))
    net.load_state_dict(sd)
    net.to(cuda)
    inp, gt = stage2_inputs(int(g['N']), int(g['L']), int(g['V']), seed=int(g['input_seed']))
    out, t, gr = _run(net, s2.MainLoss, s2.NormalLoss, inp, gt, 2, torch.from_numpy(g['nz_xyz']), cuda)
    for k in g.files:
        if k.startswith('out_'):
            assert_outputs_close(k[4:], out[k[4:]].detach().cpu(), g[k])
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)].detach()), v, 1e-4, str(k), atol=0.0)
    names, norms, projs = grad_digest(gr)
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 1e-3, 'grad norms')
    assert_close(projs, g['grad_projs'], 2e-3, 'grad projs')
    assert_close(gr['__light_dir'].cpu(), g['g_light_dir'], 1e-3, 'light dir grad')
    assert_close(gr['__light_int'].cpu(), g['g_light_int'], 1e-3, 'light int grad')


def _vis_plus_scene(h=12, w=16, P=6, Ls=(5, 4), seed=0):
    """Two in-memory views in the hand-off layout (handoff.load_view) with vis_plus tables + images / masks / lights."""
    g = torch.Generator().manual_seed(seed)
    from psnerf_amd.synthetic import look_at_pose
    views, init, imgs, omasks, ldirs = [], [], [], [], []
    for L in Ls:
        views.append({'points': torch.rand(1, h * w, 3, generator=g) * 1.2 - 0.6,
                      'normal': torch.nn.functional.normalize(torch.randn(1, h * w, 3, generator=g), dim=-1),
                      'surface_mask': torch.rand(1, h * w, generator=g) > 0.2,
                      'visibility': (torch.rand(L, h * w, generator=g) > 0.3).float(),
                      'vis_plus': (torch.rand(P, h * w, generator=g) > 0.3).float(),
                      'vis_plus_light': torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1),
                      'img_res': [h, w]})
        init.append(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1))
        imgs.append(torch.rand(L, h * w, 3, generator=g))
        omasks.append(torch.rand(h * w, generator=g) > 0.1)
        ldirs.append(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1))
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 60.0
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    poses = [look_at_pose(3.0, az_deg=10.0 * i) for i in range(len(Ls))]
    return views, init, imgs, omasks, ldirs, poses, K


def test_train_step_vis_plus_vs_oracle(cuda):
    """a24: TrainRunner.run's step body incl. light_vis_train (trainer.py:377) and the vis_plus supervision draw
    (:384-392), fed by ViewSampler.batch: HIP TrainStep vs the oracle's restatement on the same np.random stream."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.handoff import ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    from oracle import stage2 as o2
    views, init, imgs, omasks, ldirs, poses, K = _vis_plus_scene()
    sd = stage2_state_dict(o2.bear_conf(), seed=13)
    light_init = torch.cat(init, dim=0)
    NL = light_init.shape[0]
    vnum = 3
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    o_vp = dict(light=[v['vis_plus_light'] for v in views], vis=[v['vis_plus'] for v in views], view_light=init,
                view_vis=[v['visibility'] for v in views], vnum=vnum)
    ostep = o2.TrainStep(onet, o2.bear_conf(), NL, light_init, vis_plus=o_vp)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda, vis_plus=VisPlus(views, init, vnum, cuda))
    ostep.cur_iter = step.cur_iter = 5001
    for it, vidx in enumerate((1, 0)):
        results = []
        for which in ('oracle', 'hip'):
            np.random.seed(100 + it)
            ds = ViewSampler(views, imgs, omasks, ldirs, poses, K, light_bs=3, n_pixels=150)
            dev = 'cpu' if which == 'oracle' else cuda
            v, mi, gt, l_slt = ds.batch(vidx, device=dev)
            ns = int(mi['surface_mask'].sum())
            nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(5 + it)) * 0.01
            tr = ostep if which == 'oracle' else step
            terms, out = tr.step(mi, gt, l_slt, train_order=False, noise={'xyz': nz.to(dev)}, vidx=v)
            results.append((terms, out))
        (ot, oo), (pt, po) = results
        assert po['vis_train'].shape == (vnum, 150, 3)
        for k in ('sg_rgb_values', 'vis_train', 'visibility'):
            assert_outputs_close(k, po[k].detach().cpu(), oo[k].detach(), prefix='it%d ' % it)
        for k in ('total', 'vis_loss', 'sg_rgb_loss'):
            assert_close(float(pt[k]), float(ot[k]), 1e-4, '%s it%d' % (k, it), atol=0.0)
    # without a VisPlus table and without ready-made supervision lights: trainer.py:377 (V = L, supervised against
    # model_input['visibility'], loss.py:84-85)
    np.random.seed(3)
    ds = ViewSampler(views, imgs, omasks, ldirs, poses, K, light_bs=3, n_pixels=150)
    _, mi, gt, l_slt = ds.batch(0)
    nz = torch.randn(int(mi['surface_mask'].sum()), 3, generator=torch.Generator().manual_seed(9)) * 0.01
    ot, oo = ostep.step(mi, gt, l_slt, train_order=False, noise={'xyz': nz})
    pt, po = step.step({k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in mi.items()},
                       {k: v.to(cuda) for k, v in gt.items()}, l_slt.to(cuda), train_order=False, noise={'xyz': nz.to(cuda)})
    assert po['vis_train'].shape == (3, 150, 3)
    assert_outputs_close('vis_train', po['vis_train'].detach().cpu(), oo['vis_train'].detach())
    assert_close(float(pt['vis_loss']), float(ot['vis_loss']), 1e-4, 'vis_loss (trainer.py:377 branch)', atol=0.0)


def test_invalidate_packs_after_data_edit(cuda):
    """Weight packs are cached by parameter version + storage address; an edit through ``.data`` changes neither, so
    callers announce it with invalidate_packs().  load_state_dict and .to() invalidate by themselves."""
    import psnerf_amd.stage2 as s2
    net = s2.PSNetwork(s2.bear_conf()).to(cuda).eval()
    inp, _ = stage2_inputs(300, 3, 2, seed=2, device=cuda)
    with torch.no_grad():
        a = net(inp)['sg_rgb_values'].clone()
        for m in (net.albedo_net, net.visibility_net):
            m.linears[1].weight.data.mul_(1.5)  # not the first layer, no version bump
        net.invalidate_packs()
        b = net(inp)['sg_rgb_values'].clone()
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd['albedo_net.linears.2.weight'] = sd['albedo_net.linears.2.weight'] * 0.5
        net.load_state_dict(sd)
        c = net(inp)['sg_rgb_values'].clone()
    assert float((a - b).abs().max()) > 1e-4 and float((b - c).abs().max()) > 1e-4


def test_camera_rays_vs_reference_rend_util(cuda):
    """psnerf_amd.stage2.renderer.camera_rays against the outputs of the reference's own rend_util.get_camera_params."""
    from psnerf_amd.stage2.renderer import camera_rays
    g = np.load(os.path.join(GOLDEN, 'stage2_camera.npz'))
    T = lambda a: torch.from_numpy(a).to(cuda)
    rd, loc = camera_rays(T(g['uv']), T(g['pose']), T(g['K']))
    assert_close(rd.cpu(), g['ray_dirs'], 1e-6, 'ray dirs', atol=1e-7)
    assert np.array_equal(loc.cpu().numpy(), g['cam_loc'])


def test_psnetwork_normal_jitter_golden(cuda):
    """normal.net.xyz_jitter_std > 0 (renderer.py:133-140): the branch bear.conf leaves off, pinned by its own golden."""
    import psnerf_amd.stage2 as s2
    from tests.test_oracle_golden import _normal_jitter_case
    g, out, t, gr = _normal_jitter_case(s2, cuda)
    assert 'normal_jitter' in out
    for k in g.files:
        if k.startswith('out_'):
            assert_outputs_close(k[4:], out[k[4:]].detach().cpu(), g[k])
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)].detach()), v, 1e-4, str(k), atol=0.0)
    names, norms, projs = grad_digest(gr)
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 1e-3, 'grad norms')
    assert_close(projs, g['grad_projs'], 2e-3, 'grad projs')


@pytest.mark.parametrize('loss_type,phase', [('L1', 2), ('L2', 2), ('L1', 1)])
def test_fused_losses_match_modules(cuda, loss_type, phase):
    """ops.Stage2Losses (csrc/loss.hip) against MainLoss + NormalLoss of the torch formulation: every term, the total and
    the gradients of every dense output; also with normal jitter and with an empty mask."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2.loss import fused_losses
    conf = s2.bear_conf(**{'normal.net.xyz_jitter_std': 0.02})
    net = s2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(__import__('oracle.stage2', fromlist=['x']).bear_conf(**{'normal.net.xyz_jitter_std': 0.02}), seed=35))
    net.to(cuda)
    inp, gt = stage2_inputs(700, 7, 3, seed=4, device=cuda)
    lw = dict(sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1) if phase == 2 else \
        dict(sg_rgb_weight=0, albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=10)
    main, normal = s2.MainLoss(loss_type=loss_type, **lw), s2.NormalLoss(1, 0.05)
    keys = ('sg_rgb_values', 'albedo_values', 'albedo_jitter', 'rough_values', 'rough_jitter', 'vis_train', 'normal_pred', 'normal_jitter')
    res = []
    for fused_path in (False, True):
        torch.manual_seed(0)
        out = net(inp)
        leaves = {k: out[k].detach().clone().requires_grad_(True) for k in keys}
        o2 = dict(out)
        o2.update(leaves)
        count = int((o2['network_object_mask'] & o2['object_mask']).sum())
        if fused_path:
            total, t, tn = fused_losses(main, normal, o2, gt, inp, count)
        else:
            t, tn = main(o2, gt, inp, count=count), normal(o2, count=count)
            total = t['loss'] + tn['loss']
        total.backward()
        res.append((total.detach(), {k: v for k, v in list(t.items()) + [('n_' + k, v) for k, v in tn.items()] if v is not None and k != 'loss'},
                    {k: v.grad for k, v in leaves.items()}))
    (t0, d0, g0), (t1, d1, g1) = res
    assert_close(float(t1), float(t0), 2e-6, 'total', atol=0.0)
    for k in d0:
        if k in ('n_loss',):
            continue
        assert_close(float(d1[k]), float(d0[k]), 2e-6, k, atol=1e-12)
    for k in g0:
        if g0[k] is None or g1[k] is None:  # a term with weight 0: zeros in the torch formulation, no gradient at all here
            for gg in (g0[k], g1[k]):
                assert gg is None or float(gg.abs().max()) == 0.0, k
        else:
            assert_close(g1[k].cpu(), g0[k].cpu(), 2e-6, 'grad ' + k)  # max-normalised: two roundings of the same sum
    # empty mask: every term 0, no NaN
    inp2 = dict(inp)
    inp2['object_mask'] = torch.zeros_like(inp['object_mask'])
    out = net(inp2)
    total, t, tn = fused_losses(main, normal, out, gt, inp2, 0)
    assert float(total) == 0.0 and float(t['sg_rgb_loss']) == 0.0


def test_surface_idx_from_the_data_pipeline(cuda):
    """A batch that carries 'surface_idx' (the index list handoff.ViewSampler.batch builds on the host) gives bit-identical
    outputs to one whose list PSNetwork.forward derives itself with nonzero(); a TrainStep on it needs no host round trip."""
    import psnerf_amd.stage2 as s2
    net = s2.PSNetwork(s2.bear_conf()).to(cuda)
    inp, gt = stage2_inputs(900, 4, 2, seed=8, device=cuda)
    inp2, _ = stage2_inputs(900, 4, 2, seed=8, device=cuda, with_surface_idx=True)
    assert torch.equal(inp2['surface_idx'], inp['surface_mask'][0].nonzero(as_tuple=True)[0])
    nz = torch.randn(int(inp['surface_mask'].sum()), 3, device=cuda) * 0.01
    a, b = net(inp, noise={'xyz': nz}), net(inp2, noise={'xyz': nz})
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
    step = s2.TrainStep(net, s2.bear_conf(), 8, torch.nn.functional.normalize(torch.randn(8, 3), dim=-1).to(cuda), cuda)
    step.cur_iter = 5001
    t1, _ = step.step(inp2, gt, torch.arange(4, device=cuda), train_order=False, noise={'xyz': nz})
    assert bool(torch.isfinite(t1['total']))


@pytest.mark.parametrize('vis_plus,inten_train,variant', [(False, True, None), (True, True, None), (False, False, None),
                                                          (False, True, 'gtlight'), (False, True, 'fixlight'), (False, True, 'novisloss')])
def test_train_step_vs_reference_trainer_run(cuda, vis_plus, inten_train, variant):
    """a24 against the reference's OWN trainer: tests/golden/stage2_trainer.npz holds six iterations of TrainRunner.run
    (stage2/trainer.py:355-410,462-464) across the train_fix switch at iteration 5000 (:485-513), produced by calling the
    reference's methods on a duck-typed runner (tools/gen_golden.py trainer).  The HIP TrainStep replays them: loss terms of
    every iteration, final light tables, final parameters (Adam: an element whose gradient sits at the fp32 noise floor may
    step the other way, so the maximum is bounded by 2 lr per step and the bulk must agree tightly).  inten_train=False: the run
    of stage2_trainer_nointen.npz -- train.light_inten_train off as in armadillo.conf / bunny.conf (BASELINE configs[4]): no
    intensity table is trained or read, the model shades with its scalar brdf.light_intensity.  variant: the trainer switches of
    stage2/trainer.py:36-50 no shipped configuration uses -- 'gtlight' train.light_train off (the batch's lights as given, no
    tables, visibility loss on the L shading rows), 'fixlight' train.ana_fixlight (tables frozen behind the switch), 'novisloss'
    train.visibility without train.vis_loss (visibility net frozen for good; single-light layout) -- each against the reference's
    own TrainRunner.run under that switch (tests/golden/stage2_trainer_<variant>.npz)."""
    import psnerf_amd.stage2 as s2
    from tests.test_oracle_golden import _trainer_golden_steps

    def make(sd, NL, light_init, tables, over=None):
        from psnerf_amd.stage2.trainer import VisPlus
        conf = s2.bear_conf(**(over or {}))
        net = s2.PSNetwork(conf)
        net.load_state_dict(sd)
        net.to(cuda)
        vp = VisPlus(tables['views'], tables['view_light'], tables['vnum'], cuda) if tables is not None else None
        return s2.TrainStep(net, conf, NL, light_init.to(cuda), cuda, vis_plus=vp)
    g, names, logs, step = _trainer_golden_steps(make, dev=cuda, vis_plus=vis_plus, inten_train=inten_train, variant=variant)
    assert step.light_inten_train == (inten_train and variant != 'gtlight')
    for i in range(6):
        assert_close(float(logs[i]['total'].detach()), float(g['total'][i]), 1e-4 if i == 0 else 1e-3, 'it %d total' % (4998 + i), atol=0.0)
        for k, v in zip(names, g['loss_vals'][i]):
            if k == 'loss':
                continue  # the fused loss path reports the step's total under 'loss' (what the reference's dict holds after its in-place `loss += loss_normal['loss']`, trainer.py:399); the golden logged MainLoss's value before it
            got = logs[i].get(k)
            if np.isnan(v):
                assert got is None or float(got) == 0.0, (i, k, got)
            else:
                assert_close(float(got.detach()), float(v), 1e-4 if i == 0 else 1e-3, 'it %d %s' % (4998 + i, k), atol=0.0)
    # phase switch happened: BRDF nets and light tables train from iteration 5000 on
    assert step.loss.sg_rgb_weight == 1.0 and step.light_para.weight.requires_grad == (variant not in ('gtlight', 'fixlight'))
    assert any(q.requires_grad for q in step.model.visibility_net.parameters()) == (variant != 'novisloss')
    if variant in ('gtlight', 'fixlight'):  # the tables never moved
        assert torch.equal(step.light_para.weight.detach().cpu(), torch.from_numpy(g['light_init']))
    lr, lr_int, n_light_steps = 5e-4, 1e-3, 4
    d = (step.light_para.weight.detach().cpu() - torch.from_numpy(g['light_para'])).abs()
    assert float(d.max()) <= 2 * n_light_steps * lr + 1e-6 and float(d.mean()) <= 2e-5, (float(d.max()), float(d.mean()))
    d = (step.light_inten_para.weight.detach().cpu() - torch.from_numpy(g['light_inten_para'])).abs()
    assert float(d.max()) <= 2 * n_light_steps * lr_int + 1e-6 and float(d.mean()) <= 4e-5, (float(d.max()), float(d.mean()))
    for k, v in step.model.state_dict().items():
        d = (v.detach().cpu().reshape(-1)[:2048] - torch.from_numpy(g['p_' + k])).abs()
        assert float(d.max()) <= 2 * 6 * lr + 1e-6, 'param %s max diff %.3e' % (k, float(d.max()))
        assert float(d.mean()) <= 2e-5, 'param %s mean diff %.3e' % (k, float(d.mean()))
