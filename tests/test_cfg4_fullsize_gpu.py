"""BASELINE configs[4] (ARMADILLO end to end, envmap relight eval) at its STATED size, under the oracle:

  (a) `Renderer.shape_extract` + shadow-ray visibility over one full 512 x 612 view with 96 + 256 directions
      (/root/reference/stage1/shape_extract.py:112-171, stage1/model/rendering.py:297-408), a 256-pixel sub-sample of
      mask / points / normal / visibility against `oracle.stage1.Renderer(..., 'shape_extract', visibility=True)` on exactly those pixels;
  (b) the hand-off written and read back in the reference's layout, `DeviceViews` resident at 2 views x 96 lights x 313,344 px
      (+ 256 vis_plus maps each), one 32768-px x 96-light batch per view bit for bit against `ViewSampler`
      (stage2/datasets/dataset.py:137-199);
  (c) one view relit with the 16 x 32 light grid (stage2/eval.py:173-231), a pixel sub-sample against the oracle's loop.

The three share one extraction (module-scoped fixture, ~11 s of GPU time); the oracle legs run on 16 host threads (more only
oversubscribe its eager CPU kernels)."""
import numpy as np
import pytest
import torch

from tests.helpers import ATOL_DEPTH, ATOL_UNIT, assert_close, stage1_cfg, stage1_state_dict, stage2_state_dict
from psnerf_amd.synthetic import stage1_camera

pytestmark = pytest.mark.gpu

H, W = 512, 612          # the DiLiGenT-MV image
N_LIGHTS, N_PLUS = 96, 256


class _Threads(object):
    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.saved = torch.get_num_threads()
        torch.set_num_threads(min(self.n, self.saved) if self.saved > 0 else self.n)

    def __exit__(self, *exc):
        torch.set_num_threads(self.saved)
        return False


@pytest.fixture(scope='module')
def full_view():
    """The stage-1 extraction of one full view: every pixel, 96 light + 256 vis_plus directions."""
    assert torch.cuda.is_available()
    cuda = torch.device('cuda:0')
    from psnerf_amd import handoff, ops
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    cfg = stage1_cfg('bunny')   # stage1/configs/armadillo.yaml == bunny.yaml up to paths
    net = NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=11))
    ren = Renderer(net, cfg, device=cuda)
    K, c2w, S = stage1_camera(cfg, h=H, w=W)
    g = torch.Generator().manual_seed(5)
    toward = torch.nn.functional.normalize(c2w[0, :3, 3], dim=0)
    ldir = torch.nn.functional.normalize(toward[None] + 0.6 * torch.randn(N_LIGHTS, 3, generator=g), dim=-1)
    pdir = torch.nn.functional.normalize(torch.randn(N_PLUS, 3, generator=g), dim=-1)
    pdir = torch.where(((pdir * toward).sum(-1) < 0)[:, None], -pdir, pdir)   # shape_extract.py:118-131: the camera-facing hemisphere
    ops.reset_hits()
    with ops.strict():
        ex = handoff.extract_view(ren, K.to(cuda), c2w.to(cuda), S.to(cuda), H, W, light_dir=ldir.to(cuda), vis_plus_dir=pdir.to(cuda))
    torch.cuda.synchronize()
    return dict(cfg=cfg, ren=ren, K=K, c2w=c2w, S=S, ldir=ldir, pdir=pdir, ex=ex, cuda=cuda, hits=dict(ops.HITS))


def test_full_view_shape_extract_and_shadow_rays_vs_oracle(full_view):
    from oracle import stage1 as o1
    fv = full_view
    ex, cuda = fv['ex'], fv['cuda']
    n_px = H * W
    assert ex['mask'].shape == (1, n_px) and ex['points'].shape == (1, n_px, 3) and ex['normal'].shape == (1, n_px, 3)
    assert ex['visibility'].shape == (N_LIGHTS + N_PLUS, n_px)
    hit = ex['mask'][0].nonzero()[:, 0].cpu()
    assert 0.02 * n_px < hit.numel() < 0.9 * n_px, hit.numel()   # a real silhouette: neither empty nor the whole image
    # the gradient-free occupancy queries took the register-resident engines (no silent fallback at this size)
    assert fv['hits'].get('march_sweep', 0) > 0 and fv['hits'].get('shadow_indirect', 0) > 0, fv['hits']
    g = torch.Generator().manual_seed(17)
    idx = torch.cat([torch.randperm(n_px, generator=g)[:216], hit[torch.randperm(hit.numel(), generator=g)[:40]]])
    cfg = fv['cfg']
    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(stage1_state_dict(cfg, seed=11))
    with _Threads(16):
        o = o1.Renderer(onet, cfg)(ex['pixels'][:, idx.to(cuda)].cpu(), fv['K'], fv['c2w'], fv['S'], 'shape_extract', visibility=True,
                                   light_dir=torch.cat([fv['ldir'], fv['pdir']]))
    sub = idx.to(cuda)
    assert torch.equal(o['mask'], ex['mask'][:, sub].cpu()), 'hit / miss classification differs on the sub-sample'
    assert 40 <= int(o['mask'].sum()) < 256
    assert_close(ex['points'][:, sub].cpu(), o['points'], 1e-4, 'points', atol=ATOL_DEPTH)
    assert_close(ex['normal'][:, sub].cpu(), o['normal'], 1e-4, 'normal', atol=ATOL_UNIT)
    assert_close(ex['visibility'][:, sub].cpu(), o['visibility'], 1e-4, 'visibility (96 + 256 directions)', atol=ATOL_UNIT)
    # the shadow rays see something on this (seeded, perturbed) network: neither all lit nor all shadowed
    v_hit = ex['visibility'][:, ex['mask'][0]]
    assert float(v_hit.max()) > 0.2 and float(v_hit.min()) < 0.05 and float(o['visibility'][:, o['mask'][0]].std()) > 0.01


def _second_view(view0, g):
    """A second view of the same sizes with random content (the resident store and the gather are what is checked)."""
    n_px = H * W
    return {'points': torch.randn(1, n_px, 3, generator=g), 'normal': torch.randn(1, n_px, 3, generator=g),
            'surface_mask': torch.rand(1, n_px, generator=g) > 0.7, 'visibility': torch.rand(N_LIGHTS, n_px, generator=g),
            'vis_plus': torch.rand(N_PLUS, n_px, generator=g), 'vis_plus_light': torch.randn(N_PLUS, 3, generator=g), 'img_res': [H, W]}


@pytest.fixture(scope='module')
def full_store(full_view, tmp_path_factory):
    """The hand-off of the extracted view written and read back (reference npy layout) + a second view, with 8-bit images."""
    from psnerf_amd import handoff
    fv = full_view
    cuda = fv['cuda']
    out_dir = str(tmp_path_factory.mktemp('cfg4') / 'shape')
    mask = handoff.export_view(fv['ren'], fv['K'].to(cuda), fv['c2w'].to(cuda), fv['S'].to(cuda), H, W, out_dir, 1,
                               light_dir=fv['ldir'].to(cuda), vis_plus_dir=fv['pdir'].to(cuda), extracted=fv['ex'])
    assert mask.shape == (H, W)
    view0 = handoff.load_view(out_dir, 1)
    g = torch.Generator().manual_seed(23)
    views = [view0, _second_view(view0, g)]
    imgs = [torch.from_numpy(torch.randint(0, 256, (N_LIGHTS, H * W, 3), generator=g, dtype=torch.uint8).numpy().astype(np.float32) / 255.)
            for _ in range(2)]   # dataset.py:121: a decoded 8-bit PNG / 255
    omasks = [views[0]['surface_mask'][0] | (torch.rand(H * W, generator=g) > 0.9), torch.rand(H * W, generator=g) > 0.2]
    ldirs = [fv['ldir'], torch.nn.functional.normalize(torch.randn(N_LIGHTS, 3, generator=g), dim=-1)]
    poses = [fv['c2w'][0], fv['c2w'][0].clone()]
    return dict(views=views, imgs=imgs, omasks=omasks, ldirs=ldirs, poses=poses, K=fv['K'][0], mask=mask)


def test_device_views_at_the_stated_size_equal_the_host_sampler(full_view, full_store):
    from psnerf_amd.handoff import DeviceViews, ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    from tests.test_next_gpu import _same_batch
    fv, fs = full_view, full_store
    cuda = fv['cuda']
    ex = fv['ex']
    # the written-and-read-back view is the extraction, in the stage-2 (row-major) pixel order
    v0 = fs['views'][0]
    assert v0['points'].shape == (1, H * W, 3) and v0['visibility'].shape == (N_LIGHTS, H * W) and v0['vis_plus'].shape == (N_PLUS, H * W)
    assert int(v0['surface_mask'].sum()) == int(ex['mask'].sum()) and np.array_equal(v0['surface_mask'].reshape(H, W).numpy(), fs['mask'])
    init = [d + 0.05 * torch.randn(d.shape, generator=torch.Generator().manual_seed(3)) for d in fs['ldirs']]
    vp_h, vp_d = VisPlus(fs['views'], init, 8, 'cpu'), VisPlus(fs['views'], init, 8, cuda)
    host = ViewSampler(fs['views'], fs['imgs'], fs['omasks'], fs['ldirs'], fs['poses'], fs['K'], light_bs=N_LIGHTS, n_pixels=32768)
    store = DeviceViews(fs['views'], fs['imgs'], fs['omasks'], fs['ldirs'], fs['poses'], fs['K'], light_bs=N_LIGHTS, device=cuda,
                        n_pixels=32768, vis_plus=vp_d)
    assert store.tables[0]['images'].dtype == torch.uint8
    assert store.resident_bytes() > 2 * (N_LIGHTS * H * W * 3 + (N_LIGHTS + N_PLUS) * H * W * 4)   # images as bytes + both visibility sets
    np.random.seed(29)
    want = []
    for v in (0, 1):
        idx, mi, gt, l_slt = host.batch(v, device=cuda)
        mi['light_vis_train'], mi['vis_train_gt'] = (t.to(cuda) for t in vp_h.select(idx, mi['sampling_idx'][0].cpu()))
        want.append((idx, mi, gt, l_slt))
    np.random.seed(29)
    got = [store.batch(v) for v in (0, 1)]
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert a[2]['rgb'].shape == (N_LIGHTS, 32768, 3) and a[1]['vis_train_gt'].shape == (8, 32768)
        _same_batch(a, b)


def test_envmap_relight_of_a_full_view_vs_oracle(full_view, full_store):
    import psnerf_amd.stage2 as s2
    from psnerf_amd import ops
    from psnerf_amd.handoff import DeviceViews
    from psnerf_amd.stage2 import relight
    from oracle import stage2 as o2
    fv, fs = full_view, full_store
    cuda = fv['cuda']
    conf = s2.bear_conf(**{'brdf.light_intensity': 4.0})   # stage2/confs/armadillo.conf == bear.conf up to paths and the intensity
    sd = stage2_state_dict(o2.bear_conf(**{'brdf.light_intensity': 4.0}), seed=12)
    net = s2.PSNetwork(conf)
    net.load_state_dict(sd)
    net.to(cuda).eval()
    # a test-split item: every pixel of the view, no draws (dataset.py:149-151,182)
    _, mi, _, _ = DeviceViews(fs['views'], fs['imgs'], fs['omasks'], fs['ldirs'], fs['poses'], fs['K'], N_LIGHTS, cuda, n_pixels=None, split='test').batch(0)
    base = {k: mi[k] for k in ('uv', 'intrinsics', 'pose', 'object_mask', 'normal', 'points', 'surface_mask')}
    assert base['points'].shape == (1, H * W, 3)
    lh = 16
    env = np.random.RandomState(0).rand(lh, 2 * lh, 3).astype(np.float32) * (4.0 / (lh * 2 * lh))
    with ops.strict():
        rgb, vis = relight.render_envmap(net, base, env, light_h=lh, light_batch=64, visibility=True)
    torch.cuda.synchronize()
    assert rgb.shape == (H * W, 3) and vis.shape == (H * W, 3)
    live = (base['surface_mask'][0] & base['object_mask'][0]).nonzero()[:, 0]
    assert live.numel() > 0.02 * H * W
    g = torch.Generator().manual_seed(31)
    idx = torch.cat([live.cpu()[torch.randperm(live.numel(), generator=g)[:224]], torch.randperm(H * W, generator=g)[:32]])
    onet = o2.PSNetwork(o2.bear_conf(**{'brdf.light_intensity': 4.0}))
    onet.load_state_dict(sd)
    lxyz, _ = relight.gen_light_xyz(lh, 2 * lh, envmap_radius=1)
    sub = {k: (v[:, idx.to(cuda)].cpu() if v.dim() >= 2 and v.shape[1] == H * W else v.cpu()) for k, v in base.items()}
    sub['light_direction'] = torch.nn.functional.normalize(torch.from_numpy(lxyz.reshape(-1, 3)).float(), dim=-1)
    sub['light_intensity'] = torch.from_numpy(env.reshape(-1, 3))
    with _Threads(16), torch.no_grad():
        oo = onet(sub)
        ref = oo['sg_rgb_values'].sum(0).clamp(0, 1)
    assert_close(rgb[idx.to(cuda)].cpu(), ref, 1e-4, 'relit rgb (512 lights, full view, sub-sample)', atol=ATOL_UNIT)
    assert float(ref.max()) > 1e-3   # the sub-sample is lit
    # the "bf16 MFMA path" the config names, on the same view: the plain-bf16 inference engine against the exact image
    rgb16 = relight.render_envmap(net, base, env, light_h=lh, light_batch=64, precision='bf16')
    mse = float(((rgb16 - rgb)[live] ** 2).mean())
    assert -10.0 * np.log10(max(mse, 1e-20)) > 60.0, mse
