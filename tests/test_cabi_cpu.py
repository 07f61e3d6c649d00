"""No-GPU checks of the drop-in boundary: the C-ABI shared library builds/loads, exports exactly the
symbols include/psnerf_hip.h declares, and the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from tests.helpers import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'psnerf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(psn_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from psnerf_amd import build
    lib_path = build.build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), 'include/psnerf_hip.h declares %s but the library does not export it' % name
    from psnerf_amd import hip
    assert sorted(hip.SIGNATURES.keys()) == declared, 'ctypes binding and header disagree'
    assert hip.version() >= 100


def test_struct_layout_matches_header():
    from psnerf_amd import hip
    assert ctypes.sizeof(hip.PsnMlpLayer) == 40
    assert ctypes.sizeof(hip.PsnMlpDesc) == 32 + 12 * 40       # 7 x int32 (w_format since round 5), padded to 8, + the layers
    assert ctypes.sizeof(hip.PsnPackItem) == 48                # 2 pointers + int64 + 6 x int32 (format since round 5)
    assert ctypes.sizeof(hip.PsnBf16Desc) == 16 + 16          # 4 x int32 + uint8[PSN_MLP_MAX_LAYERS + 4]
    assert ctypes.sizeof(hip.PsnWnItem) == 64                  # 6 pointers + 2 x int32 + float, padded to 8


@pytest.mark.parametrize('struct', ['PsnViewBatch', 'PsnMlpDesc', 'PsnPackItem'])
def test_struct_layout_against_the_c_compiler(tmp_path, struct):
    """PsnViewBatch (the descriptor of the on-device batch assembly, psn_view_batch) mixes pointers, int64 and int fields, and
    PsnMlpDesc / PsnPackItem grew a field in round 5 (the weight-stage format): the ctypes mirrors are compared with what gcc
    makes of include/psnerf_hip.h, field by field."""
    import os, subprocess
    from psnerf_amd import hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cls = getattr(hip, struct)
    fields = [f[0] for f in cls._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "psnerf_hip.h"\nint main(void) {\n  printf("%%zu\\n", sizeof(%s));\n' % struct
                   + ''.join('  printf("%%zu\\n", offsetof(%s, %s));\n' % (struct, f) for f in fields) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[0] == ctypes.sizeof(cls)
    assert got[1:] == [getattr(cls, f).offset for f in fields]


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only check')
def test_product_path_fails_loudly_without_gpu():
    """CPU tensors must be rejected -- there is no eager/CPU fallback behind the ops."""
    from psnerf_amd import hip
    with pytest.raises(RuntimeError, match='HIP device tensor'):
        hip.composite_fwd(torch.rand(4, 8), torch.rand(4, 8, 3), True)
    with pytest.raises(RuntimeError):
        hip.pe_encode(torch.rand(4, 3), 6, 64)
    import psnerf_amd.stage2 as s2
    from psnerf_amd.synthetic import stage2_inputs
    net = s2.PSNetwork(s2.bear_conf())
    inp, _ = stage2_inputs(32, 2, 1, seed=0)
    with pytest.raises(RuntimeError):
        net(inp)


def test_argument_validation_needs_no_gpu():
    from psnerf_amd import hip
    lib = hip._lib
    rc = lib.psn_composite_fwd(None, None, 4, 8, 1, None, None, None, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    rc = lib.psn_gemm(0, 1, 4, 0, 4, 1, 4, 1, 4, 1, 4, None, 0, None, 0, None, 0, None, 0, 1, None, None, None)
    assert rc == -1 and b'bad shape' in lib.psn_last_error()
    d = hip.PsnBf16Desc()
    d.n_hidden, d.n_out = 3, 1
    rc = lib.psn_mlp_infer_bf16(ctypes.byref(d), None, None, None, 1, 1, None, 1, 1, 10, None, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    rc = lib.psn_weight_norm_fwd(0, None, None)
    assert rc == -1 and b'n_items' in lib.psn_last_error()
    rc = lib.psn_secant_step(None, 0.5, None, None, None, None, None, None, None, None, 4, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    rc = lib.psn_mlp_pack_bf16(None, 4, 4, 4, 0, 8, 0, 1, None, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    # in-kernel positional encoding: shape rules are checked before any pointer is touched
    m = hip.PsnMlpDesc()
    rc = lib.psn_mlp_infer_pe(ctypes.byref(m), None, None, None, 10, 6, 1.0, None, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    dummy = ctypes.c_void_p(64)
    m.n_layers, m.n_out, m.in_kt_a = 2, 1, 1
    rc = lib.psn_mlp_infer_pe(ctypes.byref(m), dummy, dummy, dummy, 10, 6, 1.0, dummy, None)
    assert rc == -1 and b'in_kt_a' in lib.psn_last_error()
    m.in_kt_a = 2
    rc = lib.psn_mlp_infer_pe(ctypes.byref(m), dummy, dummy, dummy, 10, 11, 1.0, dummy, None)  # 3 + 6 * 11 > 64 columns
    assert rc == -1 and b'octaves' in lib.psn_last_error()
    # weight-stage formats (PSN_W_F32 / PSN_W_BF16X2): unknown values are refused by the packer, by the in-kernel-encoding entry
    # point and by the root finder (which has no split-bf16 form)
    it = (hip.PsnPackItem * 1)()
    it[0].W, it[0].dst, it[0].ldw = 64, 64, 256
    it[0].rows, it[0].cols, it[0].transpose, it[0].n_mt, it[0].k_tiles, it[0].format = 256, 256, 0, 8, 8, 7
    rc = lib.psn_mlp_pack_layers(1, ctypes.addressof(it), None)
    assert rc == -1 and b'unknown format' in lib.psn_last_error()
    m.w_format = 5
    rc = lib.psn_mlp_infer_pe(ctypes.byref(m), dummy, dummy, dummy, 10, 6, 1.0, dummy, None)
    assert rc == -1 and b'unknown weight format' in lib.psn_last_error()
    m.w_format = hip.W_F32
    # 33..64 outputs exist for the 256-wide chain engine only; > 64 nowhere
    m.n_out = 65
    m.layers[0].init_off = m.layers[1].init_off = -1
    rc = lib.psn_mlp_infer(ctypes.byref(m), dummy, dummy, None, 1, 1, None, 1, 1, None, None, None, 0, None, None, None, None, 0, None, None, 0,
                           None, 10, dummy, None)
    assert rc == -1 and b'n_out' in lib.psn_last_error()
    # padded row sets (psn_mlp_infer_padded): a device-side count is required, the group length must be a multiple of the
    # 64-row block and save_row0 a multiple of the group length -- checked on the host, before any launch
    v = hip.PsnMlpDesc()
    v.n_layers, v.n_out, v.in_kt_a, v.in_kt_b, v.init_stride = 2, 1, 2, 0, 0
    v.layers[0].n_mt, v.layers[0].n_kt_in, v.layers[0].n_kt_act, v.layers[0].init_off, v.layers[0].b_off, v.layers[0].act = 8, 2, 0, -1, 0, hip.ACT_RELU
    v.layers[1].n_mt, v.layers[1].n_kt_in, v.layers[1].n_kt_act, v.layers[1].init_off, v.layers[1].b_off = 1, 0, 8, -1, 256
    args = (ctypes.byref(v), dummy, dummy, dummy, 1, 1000, None, 1, 1, None, None, None)
    rc = lib.psn_mlp_infer_padded(*args, 5000, 7000, dummy, None, 1000, None)
    assert rc == -1 and b'live_count' in lib.psn_last_error()
    rc = lib.psn_mlp_infer_padded(*args, 5000, 7000, dummy, dummy, 1000, None)   # 1000 is no multiple of 64
    assert rc == -1 and b'multiple of 64' in lib.psn_last_error()
    rc = lib.psn_mlp_infer_padded(*args, 5000, 7000, dummy, dummy, 1024, None)   # save_row0 = 5000 is no multiple of 1024
    assert rc == -1 and b'multiple of the period' in lib.psn_last_error()
    # on-device batch assembly: descriptor checked on the host
    rc = lib.psn_view_batch(None, None)
    assert rc == -1 and b'null' in lib.psn_last_error()
    vb = hip.PsnViewBatch()
    vb.hw, vb.width, vb.n, vb.n_lights, vb.image_type = 100, 10, 8, 2, 3
    rc = lib.psn_view_batch(ctypes.addressof(vb), None)
    assert rc == -1 and b'image_type' in lib.psn_last_error()
    vb.image_type, vb.images, vb.rgb, vb.lidx = 1, 64, 64, 64
    rc = lib.psn_view_batch(ctypes.addressof(vb), None)
    assert rc == -1 and b'value table' in lib.psn_last_error()
    vb.lut, vb.object_mask, vb.pix0 = 64, 64, 96      # identity range 96 .. 104 of a 100-pixel view
    rc = lib.psn_view_batch(ctypes.addressof(vb), None)
    assert rc == -1 and b'outside the view' in lib.psn_last_error()


def test_split_rows_backward_is_the_slice_backward():
    """ops.SplitRows == (t[:k], t[k:]) in value and gradient, incl. an unused half (gradient None -> zeros)."""
    import torch
    from psnerf_amd import ops
    g = torch.Generator().manual_seed(0)
    t = torch.randn(11, 5, generator=g, requires_grad=True)
    t2 = t.detach().clone().requires_grad_()
    wa, wb = torch.randn(4, 5, generator=g), torch.randn(7, 5, generator=g)
    a, b = ops.SplitRows.apply(t, 4)
    assert torch.equal(a, t2[:4]) and torch.equal(b, t2[4:])
    ((a * wa).sum() + (b * wb).sum() * 2).backward()
    ((t2[:4] * wa).sum() + (t2[4:] * wb).sum() * 2).backward()
    assert torch.equal(t.grad, t2.grad)
    t.grad = None
    a, b = ops.SplitRows.apply(t, 4)
    (b * wb).sum().backward()
    assert torch.equal(t.grad[:4], torch.zeros(4, 5)) and torch.equal(t.grad[4:], wb)


def test_conf_reader_on_hocon_subset():
    from psnerf_amd.stage2.conf import parse_conf, bear_conf
    text = '''
    train{
        expname = test_1
        light_bs = 10   # comment
        sg_sched_milestones = [200,400]
        visibility = True
    }
    brdf{ net{ n_freqs_xyz = 10
               xyz_jitter_std = 0.01 }
          light_intensity = 2.0 }
    '''
    c = parse_conf(text)
    assert c.get_string('train.expname') == 'test_1' and c.get_int('train.light_bs') == 10
    assert c.get_list('train.sg_sched_milestones') == [200, 400] and c.get_bool('train.visibility') is True
    assert c.get_float('brdf.net.xyz_jitter_std') == 0.01 and c.get_float('brdf.light_intensity') == 2.0
    assert c.get_int('missing.key', default=7) == 7
    with pytest.raises(KeyError):
        c.get_int('missing.key')
    assert bear_conf().get_int('visibility.net.mlp_depth') == 8


def test_state_dict_keys_match_reference_checkpoints():
    """released checkpoints must load: key names of both stages."""
    import psnerf_amd.stage1 as s1
    import psnerf_amd.stage2 as s2
    from psnerf_amd.synthetic import stage1_cfg
    k1 = set(s1.NeuralNetwork(stage1_cfg('bunny')).state_dict().keys())
    for l in range(9):
        assert {'lin%d.bias' % l, 'lin%d.weight_g' % l, 'lin%d.weight_v' % l} <= k1
    for l in range(5):
        assert {'lina%d.bias' % l, 'lina%d.weight_g' % l, 'lina%d.weight_v' % l} <= k1
    k2 = set(s2.PSNetwork(s2.bear_conf()).state_dict().keys())
    assert 'sgbasis.lobe' in k2
    for net, n in (('albedo_net', 5), ('rough_net', 3), ('normal_net', 5), ('visibility_net', 9)):
        for i in range(n):
            assert {'%s.linears.%d.weight' % (net, i), '%s.linears.%d.bias' % (net, i)} <= k2


def test_host_metrics():
    """stage2/utils/metrics.py:17-51 semantics: float64, masked, zero vectors -> 90 degrees, identical -> 100 dB."""
    import numpy as np
    from psnerf_amd.metrics import MAE, PSNR
    n1 = np.array([[0, 0, 2.0], [1, 0, 0], [0, 0, 0], [0, 1, 0]], dtype=np.float32)
    n2 = np.array([[0, 0, 1.0], [0, 1, 0], [0, 0, 1], [0, 1, 1]], dtype=np.float32)
    mean, err = MAE(n1, n2)
    # the reference divides by (norm + 1e-5): parallel vectors come out at acos(1 - 1.5e-5) = 0.31 degrees, not 0
    e0 = np.degrees(np.arccos((2 / (2 + 1e-5)) * (1 / (1 + 1e-5))))
    np.testing.assert_allclose(err, [e0, 90.0, 90.0, 45.0], atol=2e-3)
    assert abs(mean - (e0 + 225.0) / 4) < 1e-3
    mean_m, err_m = MAE(n1, n2, mask=np.array([1, 0, 0, 1]))
    assert err_m.shape == (2,) and abs(mean_m - (e0 + 45.0) / 2) < 1e-3
    img = np.full((4, 4, 3), 0.5, dtype=np.float32)
    assert PSNR(img, img) == 100
    assert abs(PSNR(img, img + 0.1) - 20.0) < 1e-4
    m = np.zeros((4, 4)); m[0, 0] = 1
    img2 = img.copy(); img2[0, 0] += 0.01
    assert abs(PSNR(img, img2, m) - 40.0) < 1e-3


def test_view_sampler_matches_reference_sampling_order(tmp_path):
    """stage2/datasets/dataset.py:137-199 on in-memory views: dictionary layout, (x, y) uv grid, light subset then
    in-mask pixel subset drawn from np.random in the reference's order."""
    import numpy as np
    import torch
    from psnerf_amd.handoff import ViewSampler
    h, w, L = 6, 8, 10
    g = torch.Generator().manual_seed(0)
    view = {'points': torch.randn(1, h * w, 3, generator=g), 'normal': torch.randn(1, h * w, 3, generator=g),
            'surface_mask': torch.rand(1, h * w, generator=g) > 0.3, 'visibility': torch.rand(L, h * w, generator=g),
            'img_res': [h, w]}
    omask = torch.rand(h * w, generator=g) > 0.25
    imgs = torch.rand(L, h * w, 3, generator=g)
    ldir = torch.randn(L, 3, generator=g)
    ds = ViewSampler([view], [imgs], [omask], [ldir], [torch.eye(4)], torch.eye(4), light_bs=4, n_pixels=12)
    np.random.seed(123)
    idx, sample, gt = ds[0]
    np.random.seed(123)
    lidx = np.random.choice(np.arange(L), 4, replace=False)
    pick = np.arange(h * w)[omask.numpy()]
    sidx = np.random.choice(pick, 12, replace=False)
    assert idx == 0 and sample['lidx'].tolist() == lidx.tolist() and sample['sampling_idx'].tolist() == sidx.tolist()
    assert bool(omask[sample['sampling_idx']].all()) and bool(sample['object_mask'].all())
    assert torch.equal(sample['light_direction'], ldir[lidx])
    assert torch.equal(sample['visibility'], view['visibility'][lidx][:, sidx])
    assert torch.equal(gt['rgb'], (imgs[lidx] * omask[None, :, None])[:, sidx])
    assert torch.equal(sample['points'], view['points'][0][sidx]) and sample['surface_mask'].shape == (12,)
    # uv = (x, y): pixel k of the row-major h*w flattening sits at x = k % w, y = k // w
    assert sample['uv'].tolist() == [[float(k % w), float(k // w)] for k in sidx.tolist()]
    # eval split / fewer lights than light_bs: all lights in order, whole image when n_pixels is None
    ds2 = ViewSampler([view], [imgs], [omask], [ldir], [torch.eye(4)], torch.eye(4), light_bs=4, split='test')
    _, s2, g2 = ds2[0]
    assert s2['lidx'].tolist() == list(range(L)) and g2['rgb'].shape == (L, h * w, 3) and 'sampling_idx' not in s2


def test_row_sparse_adam_matches_sparse_adam():
    """psnerf_amd.optim.RowSparseAdam = torch.optim.SparseAdam on the touched rows (stage2/trainer.py:126-168), with
    sparse (uncoalesced, duplicate rows) and with dense gradients; untouched rows and their moments do not move; the
    state dict has SparseAdam's layout."""
    import torch
    from psnerf_amd.optim import RowSparseAdam
    g = torch.Generator().manual_seed(0)
    n, d = 40, 3
    w0 = torch.randn(n, d, generator=g)
    ref = torch.nn.Embedding(n, d, sparse=True)
    a = torch.nn.Embedding(n, d, sparse=True)
    b = torch.nn.Embedding(n, d, sparse=False)
    for e in (ref, a, b):
        e.weight.data.copy_(w0)
    o_ref = torch.optim.SparseAdam(list(ref.parameters()), lr=5e-3)
    o_a = RowSparseAdam(list(a.parameters()), lr=5e-3)
    o_b = RowSparseAdam(list(b.parameters()), lr=5e-3)
    for it in range(6):
        rows = torch.randint(0, n, (12,), generator=g)  # with duplicates
        coef = torch.randn(12, d, generator=g)
        for e, o in ((ref, o_ref), (a, o_a), (b, o_b)):
            o.zero_grad()
            (e(rows) * coef).sum().backward()
        o_ref.step()
        o_a.step()
        o_b.step(rows=rows)
        assert torch.allclose(a.weight, ref.weight, rtol=0, atol=1e-7), it
        assert torch.allclose(b.weight, ref.weight, rtol=0, atol=1e-7), it
        untouched = torch.ones(n, dtype=torch.bool)
        untouched[rows] = False
        if it == 0:
            assert torch.equal(b.weight[untouched], w0[untouched])
    sa, sr = o_a.state_dict(), o_ref.state_dict()
    assert set(sa['state'][0].keys()) == set(sr['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert torch.allclose(sa['state'][0]['exp_avg'], sr['state'][0]['exp_avg'].to_dense() if sr['state'][0]['exp_avg'].is_sparse else sr['state'][0]['exp_avg'], atol=1e-7)
    o_a.load_state_dict(sr)  # a SparseAdam checkpoint loads
    import pytest
    with pytest.raises(RuntimeError):
        o_b.zero_grad()
        (b(torch.tensor([1])) * 1.0).sum().backward()
        o_b.step()  # dense gradient without rows


def test_vis_plus_selection_matches_reference_draw_and_layout():
    """stage2/trainer.py:384-392 (train.vis_plus): cat(P extra directions, the view's L_v initial light estimates) and
    cat(vis_plus maps, the view's stage-1 visibility maps), ONE np.random.choice(P + L_v, vnum, replace=False) per step,
    rows gathered at the step's sampling_idx -- and trainer.py:377 (light_vis_train = normalize(initial estimates[l_slt]))
    when vis_plus is off.  Also ViewSampler.batch = the collated + un-batched sample of trainer.py:364-379."""
    import numpy as np
    import torch
    from psnerf_amd.handoff import ViewSampler
    from psnerf_amd.stage2.trainer import VisPlus
    h, w, P = 5, 7, 6
    g = torch.Generator().manual_seed(0)
    Ls = [4, 3]
    views, init, imgs, omasks, ldirs = [], [], [], [], []
    for L in Ls:
        views.append({'points': torch.randn(1, h * w, 3, generator=g), 'normal': torch.randn(1, h * w, 3, generator=g),
                      'surface_mask': torch.rand(1, h * w, generator=g) > 0.3, 'visibility': torch.rand(L, h * w, generator=g),
                      'vis_plus': torch.rand(P, h * w, generator=g), 'vis_plus_light': torch.randn(P, 3, generator=g),
                      'img_res': [h, w]})
        init.append(torch.randn(L, 3, generator=g) * 2.0)  # deliberately not unit length: :388 concatenates them as they are
        imgs.append(torch.rand(L, h * w, 3, generator=g))
        omasks.append(torch.rand(h * w, generator=g) > 0.2)
        ldirs.append(torch.randn(L, 3, generator=g))
    vp = VisPlus(views, init, vnum=5, device='cpu')
    ds = ViewSampler(views, imgs, omasks, ldirs, [torch.eye(4)] * 2, torch.eye(4), light_bs=2, n_pixels=9)
    np.random.seed(7)
    vidx, mi, gt, l_slt = ds.batch(1)
    lv, gtv = vp.select(vidx, mi['sampling_idx'][0])
    # the reference's sequence on the same stream: dataset draws (lights, pixels), then the trainer's vis_plus draw
    np.random.seed(7)
    lidx = np.random.choice(np.arange(Ls[1]), 2, replace=False)
    sidx_px = np.random.choice(np.arange(h * w)[omasks[1].numpy()], 9, replace=False)
    light_plus = torch.cat([views[1]['vis_plus_light'], init[1]], dim=0)
    vis_plus_v = torch.cat([views[1]['vis_plus'].reshape(P, -1), views[1]['visibility']], dim=0)
    sidx = np.random.choice(np.arange(len(light_plus)), 5, replace=False)
    assert vidx == 1 and mi['lidx'].tolist() == lidx.tolist() and mi['sampling_idx'][0].tolist() == sidx_px.tolist()
    assert l_slt.tolist() == (Ls[0] + lidx).tolist()  # rows of the concatenated per-view light tables (trainer.py:370-373)
    assert torch.equal(lv, light_plus[sidx]) and torch.equal(gtv, vis_plus_v[sidx][:, sidx_px])
    assert mi['uv'].shape == (1, 9, 2) and mi['points'].shape == (1, 9, 3) and mi['surface_mask'].shape == (1, 9)
    assert mi['light_direction'].shape == (2, 3) and mi['visibility'].shape == (2, 9) and gt['rgb'].shape == (2, 9, 3)
    assert mi['intrinsics'].shape == (1, 4, 4) and mi['pose'].shape == (1, 4, 4)


def test_masked_losses_ignore_nonfinite_values_outside_the_mask():
    """The reference selects with boolean indexing (stage2/model/loss.py:27-38, stage1/model/losses.py:53-63): a NaN in
    a masked-OUT element must not reach the loss (a product with a 0/1 mask would turn it into NaN)."""
    import torch
    from psnerf_amd.stage2.loss import _masked_mean
    from psnerf_amd.stage1.losses import Loss
    d = torch.tensor([[[1.0, 2.0], [float('nan'), float('inf')], [3.0, 5.0]]])
    m = torch.tensor([[True, False, True]])
    assert float(_masked_mean(d, m, 2, 2)) == (1 + 2 + 3 + 5) / 4.0
    n = 6
    out = {'rgb': torch.rand(1, n, 3), 'diff_norm': torch.rand(3), 'normal_pred': torch.rand(1, n, 3)}
    ngt = torch.rand(1, n, 3)
    ngt[0, 1] = float('nan')
    nmask = torch.ones(1, n, dtype=torch.bool)
    nmask[0, 1] = False
    acc = torch.rand(1, n)
    mgt = (torch.rand(1, n) > 0.5).float()
    valid = torch.ones(1, n, dtype=torch.bool)
    valid[0, 2] = False
    t = Loss(1.0, 0.1, 0.5, 0.7)(out, torch.rand(1, n, 3), ngt, nmask, acc, mgt, valid)
    assert all(bool(torch.isfinite(v)) for v in t.values())


def test_gen_light_xyz_vs_reference_eval_utils():
    """psnerf_amd.stage2.relight.gen_light_xyz against the outputs of the reference's own utils/eval_utils.gen_light_xyz
    (16 x 32 lat-long grid of stage2/eval.py:203; fixture written by tools/gen_golden.py, which also checks split_input /
    merge_output against utils/general.py)."""
    import numpy as np
    from psnerf_amd.stage2 import relight
    from tests.helpers import GOLDEN
    import os
    g = np.load(os.path.join(GOLDEN, 'stage2_light_xyz.npz'))
    xyz, areas = relight.gen_light_xyz(16, 32, envmap_radius=1)
    assert np.array_equal(xyz, g['xyz']) and np.array_equal(areas, g['areas'])


def test_envmap_readers_known_answers_and_round_trips(tmp_path):
    """stage2/envmap_io.py (the reference's load_light / read_exr / read_hdr, eval_utils.py:11-38, without cv2):
    a hand-built run-length-encoded Radiance picture decodes to the values its RGBE bytes mean; the OpenEXR byte
    predictor + interleave is inverted correctly on a hand-built chunk; files written in every supported flavour read back;
    relight.load_light dispatches on the extension and resizes with the half-pixel bilinear rule."""
    import struct, zlib
    import numpy as np
    from psnerf_amd.stage2 import envmap_io as io
    from psnerf_amd.stage2 import relight
    # -- .hdr: 2 scanlines of 8 pixels, new-style RLE: R = run of 128, G = literals 0..7 * 16, B = run of 32, E = 129 / 120
    W = 8
    line = lambda e: bytes([2, 2, 0, W]) + bytes([128 + W, 128]) + bytes([W]) + bytes(range(0, 128, 16)) + bytes([128 + W, 32]) + bytes([128 + W, e])
    p = str(tmp_path / 'k.hdr')
    with open(p, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 8\n' + line(129) + line(120))
    a = io.read_hdr(p)
    assert a.shape == (2, 8, 3) and a.dtype == np.float32
    for y, e in ((0, 129), (1, 120)):
        s = 2.0 ** (e - 136)
        assert np.array_equal(a[y, :, 0], np.full(8, 128 * s, np.float32)) and np.array_equal(a[y, :, 2], np.full(8, 32 * s, np.float32))
        assert np.array_equal(a[y, :, 1], (np.arange(0, 128, 16) * s).astype(np.float32))
    # zero exponent = black, flat (non-RLE) pixels
    with open(p, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 2\n' + bytes([200, 100, 50, 0, 64, 128, 255, 136]))
    assert np.array_equal(io.read_hdr(p), np.array([[[0, 0, 0], [64, 128, 255]]], np.float32))
    # -- OpenEXR predictor: bytes 10 20 30 40 50 60 are stored as [10 30 50 | 20 40 60] -> deltas + 128
    enc = bytes([10, (30 - 10 + 128) & 255, (50 - 30 + 128) & 255, (20 - 50 + 128) & 255, (40 - 20 + 128) & 255, (60 - 40 + 128) & 255])
    assert io._exr_unpredict(enc) == bytes([10, 20, 30, 40, 50, 60])
    assert io._exr_unrle(bytes([2, 7, 0xFE, 1, 2]), 5) == bytes([7, 7, 7, 1, 2])  # run of 3, then 2 literals (count -2)
    # -- round trips
    g = np.random.default_rng(0)
    img = (g.random((37, 52, 3)) * 7).astype(np.float32)
    img[5:9, 3:40] = 0.5
    for comp in ('NONE', 'ZIPS', 'ZIP'):
        for half in (False, True):
            q = str(tmp_path / ('r_%s_%d.exr' % (comp, half)))
            io.write_exr(q, img, comp, half)
            ref = img.astype(np.float16).astype(np.float32) if half else img
            assert np.array_equal(io.read_exr(q), ref), (comp, half)
    for rle in (True, False):
        q = str(tmp_path / ('r_%d.hdr' % rle))
        io.write_hdr(q, img, rle=rle)
        assert np.abs(io.read_hdr(q) - img).max() <= img.max() / 128  # 8-bit mantissa under a shared exponent
    # -- load_light: extension dispatch + resize ((2 light_h, light_h), half-pixel bilinear = cv2.INTER_LINEAR)
    env = np.zeros((4, 8, 3), np.float32)
    env[:, :, 0] = np.arange(8, dtype=np.float32)[None]
    q = str(tmp_path / 'e.exr')
    io.write_exr(q, env, 'ZIP')
    small = relight.load_light(q, light_h=2)
    assert small.shape == (2, 4, 3) and np.allclose(small[0, :, 0], [0.5, 2.5, 4.5, 6.5])
    np.save(str(tmp_path / 'e.npy'), env)
    assert np.array_equal(relight.load_light(str(tmp_path / 'e.npy')), env)
    import pytest
    with pytest.raises(NotImplementedError):
        relight.load_light(str(tmp_path / 'e.png'))
