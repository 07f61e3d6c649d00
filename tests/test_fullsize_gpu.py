"""BASELINE.json's FULL benchmark sizes on the GPU, checked through size-independent properties:

* every pixel / ray is independent of the rest of the batch, so a random sub-batch evaluated on its own (different
  tile positions, different row indices into the tables, no 262144-row chunk boundary) must reproduce the
  corresponding rows of the full-size launch;
* that same sub-batch is small enough for the CPU oracle, which ties the full-size run to the reference
  semantics at the sampled rows (1e-4, the north-star tolerance);
* shading is linear in the light intensity; alpha compositing weights are a sub-partition of unity.
"""
import numpy as np
import pytest
import torch

from tests.helpers import ATOL_NORMAL, ATOL_UNIT, assert_close, assert_outputs_close, stage1_state_dict, stage2_state_dict, stage2_truth, stage1_cfg
from psnerf_amd.synthetic import stage2_inputs

pytestmark = pytest.mark.gpu

PIXEL_KEYS = ('uv', 'object_mask', 'surface_mask', 'points', 'normal', 'vis_train_gt', 'visibility')


def _sub_batch(inp, gt, idx):
    n = inp['uv'].shape[1]
    sub = {k: (v[:, idx].contiguous() if (k in PIXEL_KEYS and torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == n) else v)
           for k, v in inp.items()}
    return sub, {'rgb': gt['rgb'][:, idx].contiguous()}


def test_stage2_benchmark_size_properties(cuda):
    """configs[2]: 32768 px (90 % surface) x L = 96 shading lights x V = 8 visibility lights."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    N, L, V = 32768, 96, 8
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    inp, gt = stage2_inputs(N, L, V, seed=100)
    surf = inp['surface_mask'][0]
    ns = int(surf.sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    inp_d = {k: v.to(cuda) for k, v in inp.items()}
    gt_d = {k: v.to(cuda) for k, v in gt.items()}
    out = net(inp_d, noise={'xyz': nz.to(cuda)})
    terms = s2.MainLoss(1.0, 'L1', 0.05, 0.01, 1)(out, gt_d, inp_d)
    terms['loss'].backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert bool(torch.isfinite(out['visibility']).all()) and bool(torch.isfinite(out['sg_rgb_values']).all())

    # (1) row independence: 384 random pixels on their own
    g = torch.Generator().manual_seed(7)
    idx = torch.sort(torch.randperm(N, generator=g)[:384]).values
    surf_rank = torch.cumsum(surf.long(), 0) - 1
    nz_sub = nz[surf_rank[idx][surf[idx]]]
    sub, gt_sub = _sub_batch(inp, gt, idx)
    with torch.no_grad():
        o_sub = net({k: v.to(cuda) for k, v in sub.items()}, noise={'xyz': nz_sub.to(cuda)})
    idx_d = idx.to(cuda)
    for k in ('sg_rgb_values', 'normal_values', 'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'visibility', 'vis_train'):
        full = out[k].detach()
        full = full[:, idx_d] if full.dim() == 3 else full[idx_d]
        assert_outputs_close(k, o_sub[k].cpu(), full.cpu(), rtol=1e-5, prefix='row independence: ')

    # (2) the same sub-batch on the CPU oracle
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    with torch.no_grad():
        o_ref = onet(sub, noise={'xyz': nz_sub})
    # the same rows in float64: how far the REFERENCE arithmetic's own fp32 evaluation is from the exact value of its
    # formulas on these inputs decides what the two specular-highlight outputs are allowed (tests/helpers.py)
    truth = stage2_truth(onet, sub, noise={'xyz': nz_sub})
    for k in ('sg_rgb_values', 'normal_values', 'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'visibility', 'vis_train'):
        full = out[k].detach()
        full = full[:, idx_d] if full.dim() == 3 else full[idx_d]
        # elementwise 1e-4 |ref| + the floor of the output class (tests/helpers.py STAGE2_ATOL; the specular lobe sum
        # exp(lambda (h.n - 1)), lambda <= e^10, is floored relative to its largest value: one ulp of h.n moves the
        # sharpest lobe by 1.3e-3 relative in ANY fp32 evaluation)
        assert_outputs_close(k, full.cpu(), o_ref[k], prefix='full-size rows vs oracle: ', truth=truth)

    # (3) linearity in the light intensity (renderer.py:202-209: rgb = light_intensity * brdf * cos * vis)
    inp2 = dict(inp_d)
    inp2['light_intensity'] = inp_d['light_intensity'] * 2.0
    with torch.no_grad():
        out2 = net(inp2, noise={'xyz': nz.to(cuda)})
    m = inp_d['surface_mask'][0]
    # (rows that clamp at 1 are not linear: compare below the clamp)
    a2, a1 = out2['sg_rgb_values'][:, m].cpu(), (2.0 * out['sg_rgb_values'].detach()[:, m]).cpu()
    lin = a1 < 1.0
    assert_close(a2[lin], a1[lin], 1e-6, 'linearity', atol=1e-7)


def test_stage1_benchmark_size_properties(cuda):
    """configs[1]: 4096 rays x 128 samples (96 inner + 32 outer, it > 5000), 256 march steps: 524288 query
    points per step = two 262144-row chunks of the geometry-field chains."""
    from oracle import stage1 as o1
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    from psnerf_amd.synthetic import stage1_camera
    over = {'rendering.num_points_in': 96, 'rendering.num_points_out': 32}
    cfg = stage1_cfg('bear', **over)
    sd = stage1_state_dict(cfg, seed=11)
    net = NeuralNetwork(cfg)
    net.load_state_dict(sd)
    net.MAX_ROWS = 1 << 18  # force two chunks of the geometry-field chains (the default holds 2^20 rows per call)
    ren = Renderer(net, cfg, device=cuda)
    h, w = 96, 128
    K, c2w, S = stage1_camera(cfg, h=h, w=w)
    gen = torch.Generator().manual_seed(3)
    n_rays = 4096
    pix = torch.stack([torch.randint(0, w, (n_rays,), generator=gen).float(),
                       torch.randint(0, h, (n_rays,), generator=gen).float()], -1)[None]
    args = (K.to(cuda), c2w.to(cuda), S.to(cuda), 'unisurf')
    with torch.no_grad():
        mask = ren(pix.to(cuda), *args, add_noise=False, eval_=True, it=6000)['mask_pred'].cpu()
    n_hit = int(mask.sum())
    assert 0 < n_hit < n_rays
    nbr = torch.rand(n_hit, 3, generator=gen)
    out = ren(pix.to(cuda), *args, add_noise=False, eval_=False, it=6000, noise={'nbr': nbr.to(cuda)})
    assert np.array_equal(out['mask_pred'].cpu().numpy(), mask.numpy())
    acc = out['acc_map'].detach()
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-3  # fp32 sum of 128 weights
    loss = out['rgb'].sum() + out['diff_norm'].sum()
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())

    # row independence across the chunk boundary: rays 1984..2112 straddle query row 262144 = ray 2048 * 128
    lo, hi = 1984, 2112
    hit_rank = torch.cumsum(mask.long(), 0) - 1
    nbr_sub = nbr[hit_rank[lo:hi][mask[lo:hi]]]
    with torch.no_grad():
        o_sub = ren(pix[:, lo:hi].to(cuda), *args, add_noise=False, eval_=False, it=6000, noise={'nbr': nbr_sub.to(cuda)})
    for k in ('rgb', 'normal_pred', 'acc_map'):
        assert_close(o_sub[k].cpu(), out[k].detach()[:, lo:hi].cpu(), 1e-5, 'row independence: ' + k, atol=ATOL_UNIT)
    d_full = out['diff_norm'].detach().cpu()[hit_rank[lo:hi][mask[lo:hi]]]
    assert float((o_sub['diff_norm'].cpu() - d_full).abs().max()) < 1e-5

    # the same 128 rays on the CPU oracle
    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(sd)
    oren = o1.Renderer(onet, cfg)
    with torch.no_grad():
        o_ref = oren(pix[:, lo:hi], K, c2w, S, 'unisurf', add_noise=False, eval_=False, it=6000,
                     noise={'nbr': nbr_sub})
    assert np.array_equal(o_ref['mask_pred'].numpy(), mask[lo:hi].numpy())
    for k in ('rgb', 'normal_pred', 'acc_map'):
        assert_close(out[k].detach()[:, lo:hi].cpu(), o_ref[k], 1e-4, 'full-size rows vs oracle: ' + k, atol=ATOL_UNIT)


def test_stage1_benchmark_size_sync_free_with_jitter(cuda):
    """configs[1] through the forward the bench times: Renderer._unisurf_sync_free at 4096 rays x 128 samples with the
    stratified jitter ON (one [N, S] table + [N, 3] neighbour offsets injected).  Size-independent properties: the rows of a
    128-ray sub-batch evaluated on its own with ITS rows of the tables reproduce the full launch (row independence across
    the 262144-row chunk boundary), and the same 128 rays on the CPU oracle -- which draws per hit / miss GROUP -- with the
    tables split by the hit mask (sync_free_noise_for_reference) agree to the north-star tolerance."""
    from oracle import stage1 as o1
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    from psnerf_amd.stage1.rendering import sync_free_noise_for_reference
    from psnerf_amd.synthetic import stage1_camera
    over = {'rendering.num_points_in': 96, 'rendering.num_points_out': 32}
    cfg = stage1_cfg('bear', **over)
    sd = stage1_state_dict(cfg, seed=11)
    net = NeuralNetwork(cfg)
    net.load_state_dict(sd)
    net.MAX_ROWS = 1 << 18
    ren = Renderer(net, cfg, device=cuda)
    ren.sync_free = True
    h, w = 96, 128
    K, c2w, S = stage1_camera(cfg, h=h, w=w)
    gen = torch.Generator().manual_seed(3)
    n_rays, n_s = 4096, 128
    pix = torch.stack([torch.randint(0, w, (n_rays,), generator=gen).float(),
                       torch.randint(0, h, (n_rays,), generator=gen).float()], -1)[None]
    noise = {'full': torch.rand(n_rays, n_s, generator=gen), 'nbr_full': torch.rand(n_rays, 3, generator=gen)}
    args = (K.to(cuda), c2w.to(cuda), S.to(cuda), 'unisurf')
    out = ren(pix.to(cuda), *args, add_noise=True, eval_=False, it=6000, noise={k: v.to(cuda) for k, v in noise.items()})
    assert out.get('diff_norm_full') is not None, 'not the sync-free forward'
    mask = out['mask_pred'].cpu()
    assert 0 < int(mask.sum()) < n_rays
    acc = out['acc_map'].detach()
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-3
    (out['rgb'].sum() + torch.where(out['mask_pred'], out['diff_norm_full'], torch.zeros_like(out['diff_norm_full'])).sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())

    lo, hi = 1984, 2112  # straddles query row 262144 = ray 2048 * 128
    sub_noise = {'full': noise['full'][lo:hi], 'nbr_full': noise['nbr_full'][lo:hi]}
    with torch.no_grad():
        o_sub = ren(pix[:, lo:hi].to(cuda), *args, add_noise=True, eval_=False, it=6000,
                    noise={k: v.to(cuda) for k, v in sub_noise.items()})
    assert torch.equal(o_sub['mask_pred'].cpu(), mask[lo:hi])
    for k in ('rgb', 'normal_pred', 'acc_map'):
        assert_close(o_sub[k].cpu(), out[k].detach()[:, lo:hi].cpu(), 1e-5, 'row independence: ' + k, atol=ATOL_UNIT)
    assert float((o_sub['diff_norm_full'].cpu() - out['diff_norm_full'].detach()[lo:hi].cpu())[mask[lo:hi]].abs().max()) < 1e-5

    onet = o1.NeuralNetwork(cfg)
    onet.load_state_dict(sd)
    with torch.no_grad():
        o_ref = o1.Renderer(onet, cfg)(pix[:, lo:hi], K, c2w, S, 'unisurf', add_noise=True, eval_=False, it=6000,
                                       noise=sync_free_noise_for_reference(sub_noise, mask[lo:hi]))
    assert np.array_equal(o_ref['mask_pred'].numpy(), mask[lo:hi].numpy())
    for k in ('rgb', 'normal_pred', 'acc_map'):
        assert_close(out[k].detach()[:, lo:hi].cpu(), o_ref[k], 1e-4, 'full-size rows vs oracle: ' + k,
                     atol=ATOL_NORMAL if k == 'normal_pred' else ATOL_UNIT)
    d_full = out['diff_norm_full'].detach()[lo:hi].cpu()[mask[lo:hi]]
    assert float((d_full - o_ref['diff_norm']).abs().max()) < 1e-4


def _stage2_steps(cuda, sd, NL, light_init):
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    ostep = o2.TrainStep(onet, o2.bear_conf(), NL, light_init)
    net = s2.PSNetwork(s2.bear_conf())
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(), NL, light_init.to(cuda), cuda)
    ostep.cur_iter = step.cur_iter = 5001  # phase 2: every network and both light tables train
    return onet, ostep, net, step


def _hip_grads(step):
    gr = {k: p.grad.detach().clone() for k, p in step.model.named_parameters() if p.grad is not None}
    gr['__light_dir'] = step.light_para.weight.grad.detach().clone()
    gr['__light_int'] = step.light_inten_para.weight.grad.detach().clone()
    return gr


def test_stage2_gradients_at_the_operating_point_vs_oracle(cuda):
    """The per-rank batch of BASELINE cfg 4 -- 4096 px x L = 96 x V = 8, phase 2 -- through the product path
    (TrainStep._fwd_bwd: fused visibility pair, split-K weight gradients over K = V x Ns = 29k rows, grouped launches) against
    the CPU oracle's train step on the same weights, batch and jitter draw: EVERY parameter gradient and both light tables,
    tensor by tensor (max-normalised 1e-3: accumulation order over 3.7k - 350k rows), losses 1e-4."""
    from psnerf_amd import ops
    N, L, V, NL = 4096, 96, 8, 192
    sd = stage2_state_dict(__import__('oracle.stage2', fromlist=['x']).bear_conf(), seed=5)
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    onet, ostep, net, step = _stage2_steps(cuda, sd, NL, light_init)
    inp, gt = stage2_inputs(N, L, V, seed=100, with_surface_idx=True)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(2))[:L]
    o_inp = {k: v for k, v in inp.items() if k != 'surface_idx'}
    ot, _ = ostep.step(o_inp, gt, l_slt, train_order=False, noise={'xyz': nz})
    ops.HITS.clear()
    pt, _, _, _ = step._fwd_bwd({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt.to(cuda),
                                noise={'xyz': nz.to(cuda)})
    # the fused engines ran, the layer-wise GEMM formulation did not (a quiet fallback would still pass the numbers below)
    assert ops.HITS['VisibilityPair'] == 1 and ops.HITS['FusedReluNet'] == 3 and ops.HITS['SGShade'] == 1 \
        and ops.HITS['ScatterRows'] == 1 and ops.HITS['Stage2Losses'] == 1 and ops.HITS['LightRows'] == 1, dict(ops.HITS)
    assert ops.HITS['ReluMLP'] == 0 and ops.HITS['FusedPairMLP'] == 0, dict(ops.HITS)
    for k in ('total', 'sg_rgb_loss', 'vis_loss', 'normal_loss', 'albedo_smooth_loss', 'rough_smooth_loss'):
        assert_close(float(pt[k].detach()), float(ot[k].detach()), 1e-4, k, atol=0.0)
    gr = _hip_grads(step)
    ogr = {k: p.grad for k, p in onet.named_parameters() if p.grad is not None}
    ogr['__light_dir'] = ostep.light_para.weight.grad.to_dense()
    ogr['__light_int'] = ostep.light_inten_para.weight.grad.to_dense()
    assert sorted(gr) == sorted(ogr)
    for k in sorted(gr):
        assert_close(gr[k].cpu(), ogr[k], 1e-3, 'grad ' + k)
    # rows of the light tables that the step did not use get no gradient
    unused = torch.ones(NL, dtype=torch.bool)
    unused[l_slt] = False
    assert float(gr['__light_dir'].cpu()[unused].abs().max()) == 0.0


def test_stage2_gradient_additivity_over_pixel_shards_at_benchmark_size(cuda):
    """What pixel data parallelism relies on, at BASELINE configs[2] size (32768 px, L = 96, V = 8): with every loss term
    normalised by the GLOBAL masked-pixel count, the gradient of the full batch equals the SUM of the gradients of its 8
    pixel shards (4096 px each = the per-rank batches of cfg 4) -- every parameter and both light tables, 1e-5
    max-normalised.  The shards run through the same code path a rank runs (TrainStep._fwd_bwd with the global count)."""
    from psnerf_amd import ops
    N, L, V, NL, W = 32768, 96, 8, 192, 8
    sd = stage2_state_dict(__import__('oracle.stage2', fromlist=['x']).bear_conf(), seed=5)
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    _, _, net, step = _stage2_steps(cuda, sd, NL, light_init)
    inp, gt = stage2_inputs(N, L, V, seed=100, device=cuda)
    surf = inp['surface_mask'][0]
    ns = int(surf.sum())
    nz = (torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01).to(cuda)
    l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(2))[:L].to(cuda)
    count = (inp['surface_mask'] & inp['object_mask']).sum().float().reshape(1)
    ops.HITS.clear()
    step._fwd_bwd(inp, gt, l_slt, noise={'xyz': nz}, count=count)
    assert ops.HITS['VisibilityPair'] == 1 and ops.HITS['ReluMLP'] == 0, dict(ops.HITS)
    full = _hip_grads(step)
    surf_rank = torch.cumsum(surf.long(), 0) - 1
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    per = N // W
    for r in range(W):
        idx = torch.arange(r * per, (r + 1) * per, device=cuda)
        sub, gt_sub = _sub_batch(inp, gt, idx)
        nz_sub = nz[surf_rank[idx][surf[idx]]]
        step._fwd_bwd(sub, gt_sub, l_slt, noise={'xyz': nz_sub}, count=count)
        for k, v in _hip_grads(step).items():
            acc[k] += v.double()
    for k in sorted(full):
        assert_close(acc[k].float().cpu(), full[k].cpu(), 1e-5, 'additivity ' + k)
