"""The oracle (oracle/*.py, CPU) against golden vectors captured from the
reference itself (tools/gen_golden.py).  Runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import (COMPUTE_LOSS_CASES, GOLDEN, assert_close, compute_loss_case, grad_digest, stage1_state_dict,
                           stage2_state_dict, state_dict_digest, stage1_cfg)
from oracle import stage1 as o1
from oracle import stage2 as o2
from psnerf_amd.synthetic import stage2_inputs


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_stage1_init_path_pinned():
    g = load('stage1_net_h256.npz')
    torch.manual_seed(7)
    net = o1.NeuralNetwork(stage1_cfg('bunny'))
    assert state_dict_digest(net.state_dict()) == str(g['init_digest'])
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == 802490


@pytest.mark.parametrize('tag,over', [('h64', {'model.hidden_dim': 64, 'model.feat_size': 64}), ('h256', {})])
def test_stage1_network(tag, over):
    g = load('stage1_net_%s.npz' % tag)
    cfg = stage1_cfg('bunny', **over)
    sd = stage1_state_dict(cfg, seed=11)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = o1.NeuralNetwork(cfg)
    net.load_state_dict(sd)
    p, ray_d = T(g['p']), T(g['ray_d'])
    occ = net.infer_occ(p.clone())
    grad = net.gradient(p.clone())[:, 0]
    rgb, alpha = net(p.clone(), ray_d, return_addocc=True)
    assert_close(occ.detach(), g['occ'], 2e-6, 'occ')
    assert_close(grad.detach(), g['grad'], 2e-6, 'grad')
    assert_close(rgb.detach(), g['rgb'], 2e-6, 'rgb')
    assert_close(alpha.detach(), g['alpha'], 2e-6, 'alpha')
    assert_close(net(p, only_occupancy=True).detach(), g['occ_only'], 2e-6, 'occ_only')
    assert_close(net(p, return_logits=True).detach(), g['logits'], 2e-6, 'logits')
    loss = (rgb * T(g['c_rgb'])).sum() + (alpha * T(g['c_alpha'])).sum() + (occ * T(g['c_occ'])).sum() * 0.01 \
        + (grad * T(g['c_grad'])).sum() * 0.1
    loss.backward()
    names, norms, projs = grad_digest({k: v.grad for k, v in net.named_parameters()})
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 1e-5, 'grad norms')
    assert_close(projs, g['grad_projs'], 2e-5, 'grad projs')
    if tag == 'h64':
        for k, v in net.named_parameters():
            assert_close(v.grad, g['g_' + k], 2e-5, k)


@pytest.mark.parametrize('S', [64, 96, 128])
def test_composite(S):
    g = load('stage1_composite.npz')
    w, rgb = o1.alpha_composite(T(g['alpha%d' % S]), T(g['rgb%d' % S]))
    assert_close(w, g['w%d' % S], 1e-7, 'w')
    acc = w.sum(-1)
    assert_close(rgb + (1 - acc[:, None]), g['out%d' % S], 1e-6, 'rgb')


def test_composite_known_answers():
    a = torch.zeros(3, 16)
    w, rgb = o1.alpha_composite(a, torch.rand(3, 16, 3))
    assert float(w.abs().max()) == 0 and float(rgb.abs().max()) == 0  # alpha==0 -> rgb 0 (+1 white bg), acc 0
    a = torch.ones(3, 16)
    w = o1.alpha_composite(a)
    assert torch.allclose(w[:, 0], torch.ones(3)) and float(w[:, 1:].abs().max()) < 1e-5


def _stage1_renderer():
    cfg = stage1_cfg('bunny')
    net = o1.NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=11))
    return cfg, net, o1.Renderer(net, cfg)


def test_march_and_light_visibility():
    g = load('stage1_march.npz')
    cfg, net, ren = _stage1_renderer()
    pix, K, c2w = T(g['pix']), T(g['K']), T(g['c2w'])
    N = pix.shape[1]
    cam = o1.camera_origin(N, c2w)
    rays = o1.pixel_rays(pix, K, c2w)
    rays = rays / rays.norm(2, 2).unsqueeze(-1)
    with torch.no_grad():
        d = ren.ray_marching(cam, rays, n_steps=[256, 257], n_secant_steps=8, rad=2.0, depth_range=[2, 6])
    ref = T(g['d_i'])
    fin = torch.isfinite(ref)
    assert torch.equal(fin, torch.isfinite(d))
    assert_close(d[fin], ref[fin], 2e-6, 'd_i')
    lv = ren.light_visibility(surf=T(g['surf']), light_dir=T(g['ldir']))
    assert_close(lv, g['light_vis'], 1e-5, 'light_vis')


def test_shape_extract_vs_reference():
    """Renderer.shape_extract (rendering.py:297-376) on an int64 x-major pixel chunk with shadow-ray visibility, as
    stage1/shape_extract.py:112-139 calls it; fixture = the reference's own outputs."""
    g = load('stage1_shape_extract.npz')
    cfg, net, ren = _stage1_renderer()
    assert state_dict_digest(net.state_dict()) == str(g['sd_digest'])
    pix = T(g['pix'])
    assert pix.dtype == torch.int64
    out = ren(pix, T(g['K']), T(g['c2w']), torch.eye(4)[None], 'shape_extract', visibility=True, light_dir=T(g['ldir']))
    assert np.array_equal(out['mask'].numpy(), g['mask'])
    for k in ('normal', 'points', 'visibility'):
        assert_close(out[k], g[k], 1e-5, k)


def test_phong_renderer_vs_reference():
    """Renderer.phong_renderer (rendering.py:228-293), the third rendering_technique: fixture = the reference's own method."""
    g = load('stage1_phong.npz')
    cfg, net, ren = _stage1_renderer()
    assert state_dict_digest(net.state_dict()) == str(g['sd_digest'])
    out = ren(T(g['pix']).float(), T(g['K']), T(g['c2w']), torch.eye(4)[None], 'phong_renderer')
    assert sorted(out) == ['rgb']
    assert_close(out['rgb'], g['rgb'], 1e-5, 'rgb')
    assert np.array_equal((out['rgb'].numpy() == 1).all(-1), (g['rgb'] == 1).all(-1))


def test_arange_pixels_is_x_major_int64():
    p = o1.arange_pixels((3, 4))
    assert p.dtype == torch.int64 and p.shape == (1, 12, 2)
    assert p[0, :4].tolist() == [[0, 0], [0, 1], [0, 2], [1, 0]]  # x is the slow axis (common.py:73)


@pytest.mark.parametrize('tag', sorted(COMPUTE_LOSS_CASES))
def test_compute_loss_vs_reference_trainer(tag):
    """Trainer.compute_loss (training.py:141-198) against the reference's OWN Trainer: training mode with the normal loss,
    eval_mode (eval_=True: no smoothness term), and the mask loss (BCE of acc_map over mask_valid)."""
    g = load('stage1_compute_loss.npz')
    cfg, data, pix, noise, it, eval_mode = compute_loss_case(g, tag)
    net = o1.NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=11))
    tr = o1.Trainer(o1.Renderer(net, cfg), None, cfg)
    assert tr.n_eval_points == tr.n_training_points == 160
    terms = tr.compute_loss(data, eval_mode=eval_mode, it=it, pix=pix, noise=noise)
    assert sorted(terms) == [str(k) for k in g[tag + '_loss_names']]
    for k, v in zip(g[tag + '_loss_names'], g[tag + '_loss_vals']):
        assert_close(float(terms[str(k)]), v, 5e-5 if str(k) in ('grad_loss', 'mask_loss') else 5e-6, str(k))
    if eval_mode:
        assert float(terms['grad_loss']) == 0.0
    terms['loss'].backward()
    names, norms, projs = grad_digest({k: v.grad for k, v in net.named_parameters()})
    assert names == list(g[tag + '_grad_names'])
    assert_close(norms, g[tag + '_grad_norms'], 2e-5, 'grad norms')
    assert_close(projs, g[tag + '_grad_projs'], 1e-4, 'grad projs')


def test_compute_loss_full_image_branch_fails_like_the_reference():
    """training.py:159-165: with n_training_points >= h*w the reference walks the whole image with int64 pixel locations
    and then raises inside grid_sample (common.py:195) -- the fixture holds the reference's own exception."""
    from psnerf_amd.synthetic import stage1_batch
    g = load('stage1_compute_loss.npz')
    cfg = stage1_cfg('bunny', **{'training.n_training_points': 48})
    net = o1.NeuralNetwork(cfg)
    tr = o1.Trainer(o1.Renderer(net, cfg), None, cfg)
    with pytest.raises(RuntimeError) as e:
        tr.compute_loss(stage1_batch(cfg, h=6, w=8, seed=6), it=0)
    assert 'RuntimeError: ' + str(e.value) == str(g['full_image_error'][0])


@pytest.mark.parametrize('tag', ['it0', 'it6000', 'cfg1'])
def test_unisurf_and_loss(tag):
    """G6: the reference's own Renderer.unisurf + Loss + backward; 'cfg1' = the scale of BASELINE configs[0] (bunny.yaml, 512 rays x
    64 samples, 256 march steps, it = 0; the worst of 512 normalised surface gradients is allowed 1e-5 instead of 5e-6)."""
    g = load('stage1_unisurf_%s.npz' % tag)
    it = int(g['it'])
    cfg, net, ren = _stage1_renderer()
    noise = {'miss': T(g['nz_miss']), 'hit': T(g['nz_hit']), 'nbr': T(g['nz_nbr'])}
    out = ren(T(g['pix']), T(g['K']), T(g['c2w']), torch.eye(4)[None], 'unisurf', add_noise=True, eval_=False,
              it=it, noise=noise)
    assert np.array_equal(out['mask_pred'].numpy(), g['mask_pred'])
    for k in ('rgb', 'normal_pred', 'acc_map'):
        assert_close(out[k].detach(), g[k], 1e-5 if tag == 'cfg1' else 5e-6, k)
    assert float(np.abs(out['diff_norm'].detach().numpy() - g['diff_norm']).max()) < 5e-6
    terms = o1.Loss(1.0, 0.005, 0.05, 1.0)(out, T(g['rgb_gt']), T(g['normal_gt']), T(g['norm_mask']))
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(terms[str(k)]), v, 5e-5 if str(k) == 'grad_loss' else 5e-6, str(k))
    terms['loss'].backward()
    names, norms, projs = grad_digest({k: v.grad for k, v in net.named_parameters()})
    assert_close(norms, g['grad_norms'], 2e-5, 'grad norms')
    assert_close(projs, g['grad_projs'], 1e-4, 'grad projs')


def test_stage2_brdf():
    g = load('stage2_brdf.npz')
    b, s = o2.SGBasis(9, True)(v=T(g['v']), n=T(g['n']), l=T(g['l']), albedo=T(g['albedo']), weights=T(g['weights']))
    assert_close(b, g['brdf'], 1e-6, 'brdf')
    assert_close(s, g['spec'], 1e-6, 'spec')
    mf = o2.microfacet_brdf(T(g['l2']), T(g['v']), T(g['n']), T(g['albedo']), T(g['rough']))
    assert_close(mf, g['mf'], 1e-5, 'microfacet')
    # known answer: h == n  ->  D_k == 1 for every lobe
    n = torch.nn.functional.normalize(torch.randn(5, 3), dim=-1)
    b, s = o2.SGBasis(9, True)(v=n, n=n, l=n, albedo=torch.zeros(5, 3), weights=torch.ones(5, 27))
    assert torch.allclose(s, torch.full((5, 3), 9.0), atol=0.1)  # lambda up to e^10 amplifies 1-ulp error in h.n


@pytest.mark.parametrize('L', [1, 10])
@pytest.mark.parametrize('phase', [1, 2])
def test_stage2_psnetwork(L, phase):
    g = load('stage2_psnet_L%d_ph%d.npz' % (L, phase))
    conf = o2.bear_conf()
    sd = stage2_state_dict(conf, seed=31)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = o2.PSNetwork(conf)
    net.load_state_dict(sd)
    inp, gt = stage2_inputs(int(g['N']), L, int(g['V']), seed=int(g['input_seed']))
    if phase == 1:
        lw = dict(sg_rgb_weight=0, albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=10)
        net.albedo_net.eval().requires_grad_(False)
        net.rough_net.eval().requires_grad_(False)
    else:
        lw = dict(sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)
    ldir = inp['light_direction'].clone().requires_grad_(phase == 2)
    lint = inp['light_intensity'].clone().requires_grad_(phase == 2)
    inp['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
    inp['light_intensity'] = lint
    out = net(inp, noise={'xyz': T(g['nz_xyz'])})
    for k in g.files:
        if k.startswith('out_'):
            assert_close(out[k[4:]].detach(), g[k], 2e-6, k)
    t = dict(o2.MainLoss(loss_type='L1', **lw)(out, gt, inp))
    tn = o2.NormalLoss(1, 0.05)(out)
    t['normal_loss'] = tn['normal_loss']
    t['total'] = t['loss'] + tn['loss']
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)]), v, 2e-6, str(k))
    t['total'].backward()
    gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    if phase == 2:
        gr['__light_dir'], gr['__light_int'] = ldir.grad, lint.grad
        assert_close(ldir.grad, g['g_light_dir'], 1e-5, 'light dir grad')
        assert_close(lint.grad, g['g_light_int'], 1e-5, 'light int grad')
    names, norms, projs = grad_digest(gr)
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 2e-5, 'grad norms')
    assert_close(projs, g['grad_projs'], 1e-4, 'grad projs')


def test_stage2_psnetwork_microfacet():
    g = load('stage2_psnet_microfacet.npz')
    conf = o2.bear_conf(**{'train.render_model': 'microfacet'})
    sd = stage2_state_dict(conf, seed=33)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = o2.PSNetwork(conf)
    net.load_state_dict(sd)
    inp, gt = stage2_inputs(int(g['N']), int(g['L']), int(g['V']), seed=int(g['input_seed']))
    ldir = inp['light_direction'].clone().requires_grad_(True)
    lint = inp['light_intensity'].clone().requires_grad_(True)
    inp['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
    inp['light_intensity'] = lint
    out = net(inp, noise={'xyz': T(g['nz_xyz'])})
    for k in g.files:
        if k.startswith('out_'):
            assert_close(out[k[4:]].detach(), g[k], 2e-6, k)
    t = dict(o2.MainLoss(loss_type='L1', sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01,
                         vis_weight=1)(out, gt, inp))
    tn = o2.NormalLoss(1, 0.05)(out)
    t['normal_loss'] = tn['normal_loss']
    t['total'] = t['loss'] + tn['loss']
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)]), v, 2e-6, str(k))
    t['total'].backward()
    assert_close(ldir.grad, g['g_light_dir'], 1e-5, 'light dir grad')


def test_stage2_camera_rays_vs_reference_rend_util():
    """The fixture holds the outputs of the reference's OWN utils/rend_util.get_camera_params (tools/gen_golden.py
    imports the real module; fx != fy so both focal lengths are exercised)."""
    g = load('stage2_camera.npz')
    rd, loc = o2.camera_rays(T(g['uv']), T(g['pose']), T(g['K']))
    assert_close(rd, g['ray_dirs'], 1e-7, 'ray dirs')
    assert np.array_equal(loc.numpy(), g['cam_loc'])


def _normal_jitter_case(mod, dev='cpu'):
    g = load('stage2_psnet_normal_jitter.npz')
    conf = mod.bear_conf(**{'normal.net.xyz_jitter_std': 0.02})
    sd = stage2_state_dict(o2.bear_conf(**{'normal.net.xyz_jitter_std': 0.02}), seed=35)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = mod.PSNetwork(conf)
    net.load_state_dict(sd)
    net.to(dev)
    inp, gt = stage2_inputs(int(g['N']), int(g['L']), int(g['V']), seed=int(g['input_seed']))
    inp = {k: v.to(dev) for k, v in inp.items()}
    gt = {k: v.to(dev) for k, v in gt.items()}
    ldir = inp['light_direction'].clone().requires_grad_(True)
    inp['light_direction'] = torch.nn.functional.normalize(ldir, p=2, dim=-1)
    out = net(inp, noise={'normal': T(g['nz_normal']).to(dev), 'xyz': T(g['nz_xyz']).to(dev)})
    t = dict(mod.MainLoss(loss_type='L1', sg_rgb_weight=1.0, albedo_smooth_weight=0.05, rough_smooth_weight=0.01,
                          vis_weight=1)(out, gt, inp))
    tn = mod.NormalLoss(1, 0.05)(out)
    t.update(normal_loss=tn['normal_loss'], normal_smooth_loss=tn['normal_smooth_loss'], total=t['loss'] + tn['loss'])
    t['total'].backward()
    gr = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    return g, out, t, gr


def test_stage2_psnetwork_normal_jitter():
    """normal.net.xyz_jitter_std > 0 (renderer.py:133-140): second normal-net evaluation + NormalLoss smooth term."""
    g, out, t, gr = _normal_jitter_case(o2)
    for k in g.files:
        if k.startswith('out_'):
            assert_close(out[k[4:]].detach(), g[k], 2e-6, k)
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert_close(float(t[str(k)]), v, 2e-6, str(k))
    names, norms, projs = grad_digest(gr)
    assert names == list(g['grad_names'])
    assert_close(norms, g['grad_norms'], 2e-5, 'grad norms')
    assert_close(projs, g['grad_projs'], 1e-4, 'grad projs')


def _trainer_golden_steps(make_step, dev='cpu', vis_plus=False, inten_train=True, variant=None):
    """Replay tests/golden/stage2_trainer[_visplus].npz -- six iterations of the reference's OWN TrainRunner.run across the
    iteration-5000 train_fix switch, without / with the train.vis_plus supervision draw -- through a TrainStep built by
    ``make_step(sd, NL, light_init, tables)``; tables = None or the vis_plus tables (per view: 'vis_plus' [P, hw],
    'vis_plus_light' [P, 3], 'visibility' [L_v, hw]; + the per-view initial light estimates and vnum)."""
    # inten_train=False: stage2_trainer_nointen.npz -- train.light_inten_train off as in bunny.conf / armadillo.conf (no intensity
    # table: trainer.py:38,154-163,378-379); make_step then receives the configuration as a fifth argument
    # variant: 'gtlight' / 'fixlight' / 'novisloss' -- the trainer switches no shipped configuration uses (tools/gen_golden.py
    # TRAINER_VARIANTS: train.light_train off, train.ana_fixlight, train.visibility without train.vis_loss in the single-light layout);
    # the fixture carries its configuration overrides, make_step receives them as the fifth argument
    import json
    g = load('stage2_trainer_%s.npz' % variant if variant else
             ('stage2_trainer_visplus.npz' if vis_plus else ('stage2_trainer.npz' if inten_train else 'stage2_trainer_nointen.npz')))
    over = json.loads(str(g['overrides'])) if variant else ({} if inten_train else {'train.light_inten_train': False})
    multi_light = bool(over.get('train.multi_light', True))
    conf = o2.bear_conf()
    sd = stage2_state_dict(conf, seed=41)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    N, L, V, NL = (int(g[k]) for k in ('N', 'L', 'V', 'NL'))
    light_init = T(g['light_init'])
    tables = None
    if vis_plus:
        n0 = int(g['light_split'][0])
        views = [{'vis_plus': T(g['vp_vis'][v]).float(), 'vis_plus_light': T(g['vp_light'][v]), 'visibility': T(g['view_vis%d' % v]).float()}
                 for v in range(2)]
        tables = dict(views=views, view_light=[light_init[:n0], light_init[n0:]], vnum=int(g['vnum']))
        np.random.seed(int(g['np_seed']))
    step = make_step(sd, NL, light_init, tables, over) if over else make_step(sd, NL, light_init, tables)
    step.cur_iter = 0
    step.train_fix()                       # trainer.py:486-504: the state iteration 0 leaves behind
    step.cur_iter = int(g['first_iter'])
    names = [str(k) for k in g['loss_names']]
    logs = []
    for i in range(6):
        inp, gt = stage2_inputs(N, L, V, seed=int(g['input_seeds'][i % 3]))
        inp = {k: v.to(dev) for k, v in inp.items() if k not in ('light_intensity', 'light_vis_train', 'vis_train_gt')}
        if not multi_light:
            inp.pop('visibility')  # a data set built without train.vis_loss has none (dataset.py:168-169)
        vidx = None
        if vis_plus:
            vidx = int(g['views'][i % 3])
            inp['sampling_idx'] = T(g['sampling_idx'][i % 3])[None].to(dev)
        terms, _ = step.step(inp, {'rgb': gt['rgb'].to(dev)}, T(g['l_slt'][i]).to(dev), train_order=True,
                             noise={'xyz': T(g['noise%d' % i]).to(dev)}, vidx=vidx)
        logs.append(terms)
    return g, names, logs, step


@pytest.mark.parametrize('vis_plus,inten_train,variant', [(False, True, None), (True, True, None), (False, False, None),
                                                          (False, True, 'gtlight'), (False, True, 'fixlight'), (False, True, 'novisloss')])
def test_train_step_reproduces_the_reference_trainer_run(vis_plus, inten_train, variant):
    """a24: the oracle's TrainStep against the reference's own TrainRunner.run / train_fix (stage2/trainer.py:355-410,
    462-464, 485-513), six iterations across the switch at iteration 5000 (two with the BRDF nets and the light tables frozen
    and vis_weight 10, four with everything training), without and with the train.vis_plus draw (:384-392): every loss term of
    every iteration, the final light tables and the final network parameters.  The variants run the same six iterations under the
    trainer switches of stage2/trainer.py:36-50 that no shipped configuration uses (each fixture = the reference's own run)."""
    def make(sd, NL, light_init, tables, over=None):
        conf = o2.bear_conf(**(over or {}))
        net = o2.PSNetwork(conf)
        net.load_state_dict(sd)
        vp = None
        if tables is not None:
            vp = dict(light=[v['vis_plus_light'] for v in tables['views']], vis=[v['vis_plus'] for v in tables['views']],
                      view_light=tables['view_light'], view_vis=[v['visibility'] for v in tables['views']], vnum=tables['vnum'])
        return o2.TrainStep(net, conf, NL, light_init, vis_plus=vp)
    g, names, logs, step = _trainer_golden_steps(make, vis_plus=vis_plus, inten_train=inten_train, variant=variant)
    assert step.light_inten_train == (inten_train and variant != 'gtlight')
    assert len(step.light_optimizer.param_groups) == (2 if step.light_inten_train else 1)
    assert ('vis_loss' in names) == (variant != 'novisloss')
    for i in range(6):
        for k, v in zip(names, g['loss_vals'][i]):
            if np.isnan(v):
                assert logs[i][k] is None, (i, k)
            else:
                assert_close(float(logs[i][k]), float(v), 2e-6, 'it %d %s' % (4998 + i, k))
    assert_close(step.light_para.weight.detach(), g['light_para'], 1e-6, 'light table')
    assert_close(step.light_inten_para.weight.detach(), g['light_inten_para'], 1e-6, 'light intensity')
    sd_end = step.model.state_dict()
    _, norms, projs = grad_digest(sd_end)
    assert sorted(sd_end) == [str(k) for k in g['param_names']]
    assert_close(norms, g['param_norms'], 1e-6, 'parameter norms')
    for k, v in sd_end.items():
        assert_close(v.reshape(-1)[:2048], g['p_' + k], 1e-5, 'parameter ' + k)


def _stage1_train_step_golden(make_trainer, dev='cpu'):
    """Replay tests/golden/stage1_train_step.npz (two steps of the reference's own Trainer.train_step, training.py:46-60)."""
    from psnerf_amd.synthetic import stage1_batch
    g = load('stage1_train_step.npz')
    cfg = stage1_cfg('bunny', **{'training.n_training_points': int(g['n_points'])})
    sd = stage1_state_dict(cfg, seed=11)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    hb, wb = (int(v) for v in g['hw'])
    data = stage1_batch(cfg, h=hb, w=wb, seed=int(g['batch_seed']))
    net, tr = make_trainer(cfg, sd)
    logs = []
    for j in range(2):
        noise = {k: T(g['s%d_nz_%s' % (j, k)]).to(dev) for k in ('miss', 'hit', 'nbr')}
        logs.append(tr.train_step(data, it=int(g['s%d_it' % j]), pix=T(g['s%d_pix' % j]), noise=noise))
    return g, net, logs


def test_stage1_train_step_reproduces_the_reference_trainer():
    def make(cfg, sd):
        net = o1.NeuralNetwork(cfg)
        net.load_state_dict(sd)
        return net, o1.Trainer(o1.Renderer(net, cfg), torch.optim.Adam(net.parameters(), lr=1e-4), cfg)
    g, net, logs = _stage1_train_step_golden(make)
    for j in range(2):
        for k, v in zip(g['s%d_loss_names' % j], g['s%d_loss_vals' % j]):
            assert_close(float(logs[j][str(k)]), float(v), (5e-5 if str(k) == 'grad_loss' else 5e-6) * (1 if j == 0 else 50), '%s step %d' % (k, j))
    for k, v in net.state_dict().items():  # Adam: sign flips at the fp32 noise floor move an element by up to 2 lr per step
        d = (v.reshape(-1)[:1024] - T(g['p_' + k])).abs()
        assert float(d.max()) <= 2 * 2 * 1e-4 + 1e-6 and float(d.mean()) <= 1e-5, (k, float(d.max()), float(d.mean()))


def test_relight_loop_vs_reference_pieces():
    """The oracle network under the relighting loop of eval.py:173-218 (fixture assembled from the reference's own pieces)."""
    g = load('stage2_relight.npz')
    conf = o2.bear_conf()
    net = o2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=12))
    hr, wr = (int(v) for v in g['hw'])
    inp, _ = stage2_inputs(hr * wr, 1, 1, seed=int(g['input_seed']), h=hr, w=wr)
    from psnerf_amd.stage2.relight import gen_light_xyz
    lxyz = gen_light_xyz(int(g['light_h']), 2 * int(g['light_h']), envmap_radius=1)[0].reshape(-1, 3)
    mi = {'object_mask': torch.ones(1, hr * wr, dtype=torch.bool), 'uv': T(g['uv'])[None], 'intrinsics': inp['intrinsics'], 'pose': inp['pose'],
          'normal': torch.ones(1, hr * wr, 3), 'points': inp['points'], 'surface_mask': inp['surface_mask'],
          'light_direction': torch.nn.functional.normalize(torch.tensor(lxyz).float(), p=2, dim=-1),
          'light_intensity': T(g['env']).reshape(-1, 3)}
    with torch.no_grad():
        out = net(mi)
    assert_close(out['sg_rgb_values'].sum(0).clamp(0, 1).reshape(hr, wr, 3), g['rgb'], 5e-6, 'relit rgb')
    assert_close(out['visibility'].mean(0).reshape(hr, wr, 3), g['visibility'], 5e-6, 'visibility')


def _eval_view_case(net_cls, conf_fn, dev='cpu'):
    """tests/golden/stage2_eval_view.npz: the test-view render of stage2/eval.py:314-417 and its material-edit variant (:233-312),
    assembled from the reference's own pieces (tools/gen_golden.py) -> (golden, model, model_input, lights, intensities)."""
    from psnerf_amd.stage2 import relight
    g = load('stage2_eval_view.npz')
    conf = conf_fn(**{'brdf.net.xyz_jitter_std': 0, 'normal.net.xyz_jitter_std': 0})
    sd = stage2_state_dict(o2.bear_conf(**{'brdf.net.xyz_jitter_std': 0, 'normal.net.xyz_jitter_std': 0}), seed=14)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = net_cls(conf)
    net.load_state_dict(sd)
    net.to(dev).eval()
    hv, wv = (int(v) for v in g['hw'])
    inp, _ = stage2_inputs(hv * wv, 1, 1, seed=int(g['input_seed']), h=hv, w=wv)
    mi = {'object_mask': torch.ones(1, hv * wv, dtype=torch.bool), 'uv': T(g['uv'])[None], 'intrinsics': inp['intrinsics'], 'pose': inp['pose'],
          'normal': torch.ones(1, hv * wv, 3), 'points': inp['points'], 'surface_mask': inp['surface_mask']}
    mi = {k: v.to(dev) for k, v in mi.items()}
    ld, li = relight.eval_lights(None, T(g['lidx']).long(), T(g['light_para']).to(dev), T(g['light_inten']).to(dev), light_offset=int(g['light_offset']))
    return g, net, mi, ld, li


@pytest.mark.parametrize('edit', [False, True])
def test_render_view_loop_vs_reference_pieces(edit):
    """relight.render_view (the product's loop over light batches and pixel chunks + the per-view maps of eval.py:371-406) with the
    oracle network as the model, against the reference's own PSNetwork + split_input / merge_output in eval.py's order; edit: the
    material-edit variant with relight.edit_material's albedo_new / basis_new (eval.py:121-139, renderer.py:170-183)."""
    from psnerf_amd.stage2 import relight
    g, net, mi, ld, li = _eval_view_case(o2.PSNetwork, o2.bear_conf)
    kw, tag = {}, 'view_'
    if edit:
        an, bn, name = relight.edit_material(color=str(g['color']), basis=int(g['basis']), edit_albedo=True, edit_specular=True)
        assert name == '#4080c0_sg4' and an.dtype == np.float32
        kw, tag = {'albedo_new': an, 'basis_new': bn}, 'edit_'
    hv, wv = (int(v) for v in g['hw'])
    maps = relight.render_view(net, mi, ld, li, light_batch=int(g['light_batch']), pixel_chunk=1024, **kw)
    whole = relight.render_view(net, mi, ld, li, light_batch=64, **kw)   # one light batch, one pixel chunk: the same maps
    for k in ('rgb', 'rough', 'visibility'):
        assert_close(maps[k].reshape(-1, hv, wv, 3), g[tag + k], 5e-6, k)
        assert_close(whole[k], maps[k], 5e-6, k + ' (unchunked)')
    for k in ('normal', 'albedo'):
        assert_close(maps[k].reshape(hv, wv, 3), g[tag + k], 5e-6, k)
    assert np.array_equal(maps['mask'].reshape(hv, wv).numpy(), g[tag + 'mask'].astype(bool))
    if edit:  # every surface pixel carries the new albedo; the single lobe's weight makes the specular colour light-dependent only
        m = maps['mask']
        assert_close(maps['albedo'][m], np.broadcast_to(kw['albedo_new'], (int(m.sum()), 3)), 1e-7, 'edited albedo')
