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
