"""bf16 inference engine (csrc/mlp_infer_bf16.hip; BASELINE config 5 "bf16 MFMA path ... envmap relight eval").

Not a parity-gated path: the reference computes in fp32 and every training / parity test uses the exact fp32 engine.
The checks here pin the bf16 engine (a) against a float64 emulation of exactly its roundings (weights, inputs and
post-ReLU activations rounded to bf16 RNE, bias as bf16 hi + lo, wide accumulation) -- the only differences left are
the fp32 accumulation order and the rounding flips it causes (one bf16 ulp of one activation), tolerance 3e-3 on
O(0.3) logits -- and (b) against the fp32 path at image level: relit images within 0.05 dB PSNR of the fp32 render
measured against the same ground truth (north star: "rendered PSNR within 0.05 dB")."""
import numpy as np
import pytest
import torch

from psnerf_amd.synthetic import stage2_inputs
from psnerf_amd import metrics
from tests.helpers import stage2_state_dict

pytestmark = pytest.mark.gpu


def _net(din_a, din_b, depth, skip_at, n_out, seed, dev):
    g = torch.Generator().manual_seed(seed)
    din = din_a + din_b
    Ws, bs = [], []
    for li in range(depth):
        fan_in = (din if li == 0 else 256) + (din if li - 1 == skip_at else 0)
        out = 256 if li < depth - 1 else n_out
        k = 1.0 / fan_in ** 0.5
        Ws.append(((torch.rand(out, fan_in, generator=g) * 2 - 1) * k * 1.7).to(dev))
        bs.append(((torch.rand(out, generator=g) * 2 - 1) * k).to(dev))
    return Ws, bs


def _emulate(Ws, bs, xa, xb, skip_at, out_act):
    r = lambda t: t.to(torch.bfloat16).double()
    x = r(xa) if xb is None else torch.cat([r(xa), r(xb)], dim=1)
    h = None
    for li in range(len(Ws) - 1):
        bh = bs[li].to(torch.bfloat16).float()
        b = bh.double() + r(bs[li] - bh)
        inp = x if li == 0 else (torch.cat([h, x], dim=1) if li - 1 == skip_at else h)
        h = r(torch.relu(inp @ r(Ws[li]).t() + b).float())
    out = h @ r(Ws[-1]).t() + bs[-1].double()
    if out_act == 1:
        out = torch.sigmoid(out)
    elif out_act == 2:
        out = torch.sigmoid(-10.0 * out)
    return out.float()


def _table(n, d, seed, dev):
    t = torch.zeros(n, 64, device=dev)
    t[:, :d] = torch.randn(n, d, generator=torch.Generator().manual_seed(seed)).clamp(-1, 1).to(dev)
    return t


@pytest.mark.parametrize('nA,nB,depth,skip_at,n_out,out_act', [
    (1000, 7, 8, 3, 1, 0),     # visibility_net shape of bear.conf, ragged tail (7000 rows = 27 workgroups + 88 rows)
    (1000, 7, 8, 3, 1, 1),
    (37, 3, 8, 3, 1, 0),       # less than one workgroup
    (513, 2, 5, 1, 3, 0),      # other depth / skip position, several outputs
    (300, 4, 4, -100, 32, 2),  # no skip connection, full final tile, occupancy output
    (256, 1, 2, -100, 1, 0),   # one hidden layer, exactly one workgroup
    (700, 3, 12, 5, 2, 0),     # the deepest network both engines hold (11 hidden layers + final)
])
def test_bf16_engine_vs_emulation(cuda, nA, nB, depth, skip_at, n_out, out_act):
    from psnerf_amd import fused
    Ws, bs = _net(63, 63, depth, skip_at, n_out, seed=depth + n_out, dev=cuda)
    ta, tb = _table(nA, 63, 1, cuda), _table(nB, 63, 2, cuda)
    pk = fused.pack_relu_mlp_bf16(Ws, bs, 63, 63, skip_at, out_act)
    out = pk(ta.to(torch.bfloat16), nA * nB, a_div=1, a_mod=nA, tab_b=tb.to(torch.bfloat16), b_div=nA, b_mod=nB)
    ref = _emulate(Ws, bs, ta[:, :63].tile(nB, 1), tb[:, :63].repeat_interleave(nA, dim=0), skip_at, out_act)
    assert out.shape == (nA * nB, n_out)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 3e-3
    # and it is a bf16-accurate evaluation of the fp32 network
    p32 = fused.pack_relu_mlp(Ws, bs, 63, 63, skip_at, out_act)
    o32 = p32(ta, nA * nB, a_div=1, a_mod=nA, tab_b=tb, b_div=nA, b_mod=nB)
    assert (out - o32).abs().max().item() < 2e-2


def _emulate_grouped(Ws, bs, xa, xb, skip_at, out_act, din_a):
    """The grouped form: table-A columns and activations rounded to bf16 as before; the group's part of an input layer,
    W_b x_g + b, is an fp32 product entering as bias (bf16 hi + lo)."""
    r = lambda t: t.to(torch.bfloat16).double()
    h = None
    for li in range(len(Ws) - 1):
        W = Ws[li]
        if li == 0 or li - 1 == skip_at:
            o = 0 if li == 0 else 256
            Wa, Wb = W[:, o:o + din_a], W[:, o + din_a:]
            bias = (xb @ Wb.t() + bs[li]).float()
            bh = bias.to(torch.bfloat16).float()
            z = r(xa) @ r(Wa).t() + bh.double() + r(bias - bh)
            if li > 0:
                z = z + h @ r(W[:, :256]).t()
        else:
            bh = bs[li].to(torch.bfloat16).float()
            z = h @ r(W).t() + bh.double() + r(bs[li] - bh)
        h = r(torch.relu(z).float())
    out = h @ r(Ws[-1]).t() + bs[-1].double()
    if out_act == 1:
        out = torch.sigmoid(out)
    elif out_act == 2:
        out = torch.sigmoid(-10.0 * out)
    return out.float()


@pytest.mark.parametrize('nA,nB,depth,skip_at,n_out,out_act', [
    (1000, 7, 8, 3, 1, 1),     # visibility_net shape of bear.conf: 4 workgroups per group, the last one ragged (232 rows)
    (37, 3, 8, 3, 1, 0),       # less than one workgroup per group
    (256, 2, 2, -100, 1, 0),   # one hidden layer, groups of exactly one workgroup
    (513, 5, 5, 1, 3, 0),      # other depth / skip position, several outputs
    (300, 4, 12, 5, 32, 2),    # deepest network, full final tile
])
def test_bf16_engine_grouped_vs_emulation(cuda, nA, nB, depth, skip_at, n_out, out_act):
    """psn_mlp_infer_bf16_grouped: group-major rows, the group's half of the input block folded into a per-group bias."""
    from psnerf_amd import fused
    Ws, bs = _net(63, 63, depth, skip_at, n_out, seed=depth + n_out, dev=cuda)
    ta, tb = _table(nA, 63, 1, cuda), _table(nB, 63, 2, cuda)
    pk = fused.pack_relu_mlp_bf16_grouped(Ws, bs, 63, 63, skip_at, out_act)
    out = pk(ta.to(torch.bfloat16), tb)
    ref = _emulate_grouped(Ws, bs, ta[:, :63].tile(nB, 1), tb[:, :63].repeat_interleave(nA, dim=0), skip_at, out_act, 63)
    assert out.shape == (nA * nB, n_out)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 3e-3
    # closer to the fp32 network than the two-table form (the group's features and weights are not rounded)
    p32 = fused.pack_relu_mlp(Ws, bs, 63, 63, skip_at, out_act)
    o32 = p32(ta, nA * nB, a_div=1, a_mod=nA, tab_b=tb, b_div=nA, b_mod=nB)
    assert (out - o32).abs().max().item() < 2e-2
    with pytest.raises(RuntimeError):
        pk(ta, tb)  # fp32 table A


def test_bf16_engine_single_table_and_index_maps(cuda):
    from psnerf_amd import fused
    Ws, bs = _net(40, 0, 6, 2, 2, seed=5, dev=cuda)
    ta = _table(90, 40, 3, cuda)
    pk = fused.pack_relu_mlp_bf16(Ws, bs, 40, 0, 2, 0)
    n_rows = 90 * 5 + 13
    out = pk(ta.to(torch.bfloat16), n_rows, a_div=5, a_mod=90)  # row q reads table row (q // 5) % 90
    idx = (torch.arange(n_rows, device=cuda) // 5) % 90
    ref = _emulate(Ws, bs, ta[idx, :40], None, 2, 0)
    assert (out - ref).abs().max().item() < 3e-3


def test_bf16_engine_rejects_bad_arguments(cuda):
    from psnerf_amd import fused
    Ws, bs = _net(63, 63, 4, 1, 1, seed=1, dev=cuda)
    pk = fused.pack_relu_mlp_bf16(Ws, bs, 63, 63, 1, 0)
    with pytest.raises(RuntimeError):
        pk(torch.zeros(10, 64, device=cuda), 10)  # fp32 table
    with pytest.raises(RuntimeError):
        pk(torch.zeros(10, 32, device=cuda, dtype=torch.bfloat16), 10)  # wrong table width


def test_relight_bf16_psnr_parity(cuda):
    """Envmap relighting (stage2/eval.py:199-218) with visibility_net on the bf16 engine: PSNR against a ground-truth
    image within 0.05 dB of the fp32 render, and the two renders > 45 dB apart from each other."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2 import relight
    conf = s2.bear_conf()
    net = s2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=12))
    net.to(cuda).eval()
    N, lh = 4096, 8
    inp, _ = stage2_inputs(N, 1, 1, seed=3)
    base = {k: inp[k].to(cuda) for k in ('uv', 'intrinsics', 'pose', 'object_mask', 'normal', 'points', 'surface_mask')}
    env = np.random.RandomState(0).rand(lh, 2 * lh, 3).astype(np.float32) * (4.0 / (lh * 2 * lh))
    rgb32, vis32 = relight.render_envmap(net, base, env, light_h=lh, light_batch=64, visibility=True)
    rgb16, vis16 = relight.render_envmap(net, base, env, light_h=lh, light_batch=64, visibility=True, precision='bf16')
    assert net.inference_precision == 'fp32'  # restored
    assert not torch.equal(rgb32, rgb16)       # the bf16 engine really ran
    gt = (rgb32 + 0.03 * torch.randn(rgb32.shape, generator=torch.Generator().manual_seed(0)).to(cuda)).clamp(0, 1)
    np_ = lambda t: t.cpu().numpy()
    p32 = metrics.PSNR(np_(rgb32), np_(gt))
    p16 = metrics.PSNR(np_(rgb16), np_(gt))
    assert abs(p32 - p16) < 0.05, (p32, p16)
    assert metrics.PSNR(np_(rgb16), np_(rgb32)) > 45.0
    assert (vis16 - vis32).abs().max().item() < 2e-2


def test_bf16_never_used_with_gradients(cuda):
    """inference_precision only affects gradient-free evaluations: a training forward is bit-identical."""
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    net = s2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=12))
    net.to(cuda).train()
    inp, _ = stage2_inputs(512, 4, 2, seed=3)
    inp = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in inp.items()}
    torch.manual_seed(0)
    a = net(inp)['sg_rgb_values']
    net.inference_precision = 'bf16'
    torch.manual_seed(0)
    b = net(inp)['sg_rgb_values']
    assert torch.equal(a, b)


def test_train_vis_bf16_option(cuda):
    """conf train.vis_bf16 (opt-in, default off): only the detached L shading rows move to the bf16 engine; the
    supervised visibility rows, every gradient path and the loss structure are unchanged, so the loss terms stay
    within 1e-4 relative of the fp32 step and vis_loss is bit-identical."""
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    sd = stage2_state_dict(conf, seed=12)
    inp, gt = stage2_inputs(1024, 6, 3, seed=3)
    inp = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in inp.items()}
    gt = {k: v.to(cuda) for k, v in gt.items()}
    res = {}
    for flag in (False, True):
        net = s2.PSNetwork(conf)
        net.load_state_dict(sd)
        net.to(cuda).train()
        assert net.train_vis_bf16 is False  # default
        net.train_vis_bf16 = flag
        torch.manual_seed(0)
        out = net(inp)
        loss = s2.MainLoss(1.0, 'L1', 0.01, 0.01, 1.0)(out, gt, inp)
        loss['loss'].backward()
        g = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
        res[flag] = (out, {k: float(v.detach()) for k, v in loss.items()}, g)
    (o0, l0, g0), (o1, l1, g1) = res[False], res[True]
    assert not torch.equal(o0['visibility'], o1['visibility'])      # the bf16 engine ran for the shading rows
    assert torch.equal(o0['vis_train'], o1['vis_train'])            # supervised rows: exact fp32 path
    assert l0['vis_loss'] == l1['vis_loss']
    assert abs(l0['sg_rgb_loss'] - l1['sg_rgb_loss']) <= 1e-4 * abs(l0['sg_rgb_loss'])
    assert float((g1 - g0).norm() / g0.norm()) < 1e-3


def _ref64(Ws, bs, xa, xb_rows, skip_at, dtype):
    """The network in plain torch at ``dtype`` on the light-major rows (g, n): input [xa[n] | xb[g]]."""
    Ws, bs = [w.to(dtype) for w in Ws], [b.to(dtype) for b in bs]
    x = torch.cat([xa.to(dtype).repeat(xb_rows.shape[0], 1), xb_rows.to(dtype).repeat_interleave(xa.shape[0], dim=0)], dim=1)
    h = None
    for li in range(len(Ws) - 1):
        inp = x if li == 0 else (torch.cat([h, x], dim=1) if li - 1 == skip_at else h)
        h = torch.relu(inp @ Ws[li].t() + bs[li])
    return h @ Ws[-1].t() + bs[-1]


@pytest.mark.parametrize('nA,nB,depth,skip_at,n_out', [(1000, 7, 9, 4, 1), (129, 3, 9, 4, 1), (37, 2, 5, 1, 3), (4096, 12, 9, 4, 1)])
def test_split_bf16_engine_has_fp32_class_accuracy(cuda, nA, nB, depth, skip_at, n_out):
    """csrc/mlp_infer_x3.hip (every operand as three bf16 planes, six partial products per multiply): its distance from the
    float64 evaluation of the network is that of an fp32 evaluation (compared with torch's own fp32 result on the same rows),
    and it meets the elementwise parity bound of the exact-fp32 path, 1e-4 |ref| + 1e-5 -- which the plain bf16 engine misses
    by two orders of magnitude."""
    from psnerf_amd import fused
    Ws, bs = _net(63, 63, depth, skip_at, n_out, 5, cuda)
    xa, xb = _table(nA, 63, 1, cuda), _table(nB, 63, 2, cuda)
    pk = fused.pack_relu_mlp_x3_grouped(Ws, bs, 63, 63, skip_at)
    got = pk(xa, xb)
    ref64 = _ref64(Ws, bs, xa[:, :63], xb[:, :63], skip_at, torch.float64)
    ref32 = _ref64(Ws, bs, xa[:, :63], xb[:, :63], skip_at, torch.float32)
    assert got.shape == ref64.shape == (nA * nB, n_out)
    e_x3 = (got.double() - ref64).abs()
    e_32 = (ref32.double() - ref64).abs()
    scale = float(ref64.abs().max())
    print('max |x3 - f64| %.3e, max |torch fp32 - f64| %.3e, output scale %.3f' % (float(e_x3.max()), float(e_32.max()), scale))
    assert float(e_x3.max()) <= 4.0 * float(e_32.max()) + 1e-7 * scale, (float(e_x3.max()), float(e_32.max()))
    assert float(e_x3.mean()) <= 3.0 * float(e_32.mean()) + 1e-8 * scale
    bound = 1e-4 * ref64.abs() + 1e-5
    assert bool((e_x3 <= bound).all())
    # the plain bf16 engine on the same rows, for scale
    pk16 = fused.pack_relu_mlp_bf16_grouped(Ws, bs, 63, 63, skip_at)
    e_16 = (pk16(xa.to(torch.bfloat16), xb).double() - ref64).abs()
    assert float(e_16.max()) > 30.0 * float(e_x3.max())


@pytest.mark.parametrize('flag_name,bound', [('train.vis_bf16x6', 2e-6), ('train.vis_bf16x3', 1e-4)])
@pytest.mark.parametrize('N,L,V', [(3000, 5, 8), (777, 96, 3)])
def test_train_vis_bf16x6_passes_the_exact_fp32_parity_gate(cuda, N, L, V, flag_name, bound):
    """conf train.vis_bf16x6 (opt-in experiment): the L shading rows of a training forward on the split-bf16 engine -- or, conf
    train.vis_bf16x3, through the exact engine's own kernel on split-bf16 weight stages (three partial products).  Gate =
    the SAME checks as the exact-fp32 path (tests/test_stage2_gpu.py::test_psnetwork_vs_oracle): every output elementwise
    within 1e-4 |ref| + floor of the CPU oracle (the two specular outputs by the measured float64 allowance), loss terms 1e-4,
    every parameter gradient 1e-3; the supervised rows stay bit-identical to the fp32 engine's."""
    import psnerf_amd.stage2 as s2
    from oracle import stage2 as o2
    from tests.helpers import assert_close, assert_outputs_close
    from tests.test_stage2_gpu import _run, _truth
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    inp, gt = stage2_inputs(N, L, V, seed=N)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    o_out, o_t, o_g = _run(onet, o2.MainLoss, o2.NormalLoss, inp, gt, 2, nz, 'cpu')
    truth = _truth(o2.bear_conf(), sd, inp, nz)
    outs = {}
    for flag in (False, True):
        net = s2.PSNetwork(s2.bear_conf(**{flag_name: flag}))
        assert getattr(net, flag_name.replace('.', '_')) is flag
        net.load_state_dict(sd)
        net.to(cuda)
        outs[flag] = _run(net, s2.MainLoss, s2.NormalLoss, inp, gt, 2, nz, cuda)
    out, t, gr = outs[True]
    for k in o_out:
        if torch.is_tensor(o_out[k]) and o_out[k].dtype.is_floating_point:
            assert_outputs_close(k, out[k].detach().cpu(), o_out[k].detach(), truth=truth)
    for k in o_t:
        if o_t[k] is not None:
            assert_close(float(t[k]), float(o_t[k]), 1e-4, k, atol=0.0)
    assert sorted(gr.keys()) == sorted(o_g.keys())
    for k in o_g:
        assert_close(gr[k].cpu(), o_g[k], 1e-3, 'grad ' + k)
    out0 = outs[False][0]
    assert torch.equal(out['vis_train'], out0['vis_train'])                   # supervised rows: the exact fp32 engine
    assert not torch.equal(out['visibility'], out0['visibility'])             # the split engine ran for the shading rows
    m = inp['surface_mask'][0].to(cuda)
    d = (out['visibility'] - out0['visibility'])[:, m].abs().max()
    assert float(d) < bound, 'split engine vs fp32 engine on the shading rows: %.3e' % float(d)


def test_relight_bf16x6_matches_fp32_render(cuda):
    """inference_precision = 'bf16x6' for a gradient-free evaluation (relighting): pixel-level agreement with the fp32 render."""
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    net = s2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=12))
    net.to(cuda).eval()
    inp, _ = stage2_inputs(2000, 16, 1, seed=4, device=cuda)
    res = {}
    with torch.no_grad():
        for prec in ('fp32', 'bf16x6', 'bf16x3', 'bf16'):
            net.inference_precision = prec
            res[prec] = net(inp)['sg_rgb_values'].clone()
    net.inference_precision = 'fp32'
    d6 = float((res['bf16x6'] - res['fp32']).abs().max())
    d3 = float((res['bf16x3'] - res['fp32']).abs().max())
    d1 = float((res['bf16'] - res['fp32']).abs().max())
    assert d6 < 5e-6 and d1 > 20 * d6, (d6, d1)
    assert d6 < d3 < 1e-4 and d1 > 5 * d3, (d6, d3, d1)   # (three partial products: between the six-product engine and plain bf16)


@pytest.mark.parametrize('K,M,N,seg2', [(40000, 256, 256, True), (7001, 217, 256, False), (33333, 256, 200, True), (300, 256, 256, False),
                                        (65536, 129, 131, False)])
def test_split_bf16_weight_gradient_gemm_has_fp32_class_accuracy(cuda, K, M, N, seg2):
    """psn_gemm_tn_grouped_x3 (the 256 x 256-tile weight-gradient products dW = dZ^T X (+ U^T dR) of stage1/model/network.py:85-106
    on the bf16 matrix pipe: three bf16 planes per operand, six partial products, fp32 accumulation) against float64: the error
    of the exact fp32 kernel on the same operands is the yardstick (the split kernel may not be more than 1.5x worse), and both
    stay below 2e-6 relative; ragged K (clamped + zeroed tail rows), narrow operands as column slices of 256-wide dumps, the
    second product segment, the column sums (bias gradients)."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(K + M)
    its = []
    for _ in range(2):
        it = dict(A=(torch.randn(K, 256, generator=g) * 0.3).to(cuda)[:, :M], B=(torch.randn(K, 256, generator=g).abs() * 0.1).to(cuda)[:, :N], colsum=True)
        if seg2:
            it['A2'], it['B2'] = torch.randn(K, 256, generator=g).to(cuda)[:, :M], (torch.randn(K, 256, generator=g) * 0.05).to(cuda)[:, :N]
        its.append(it)
    ref = []
    for it in its:
        r = it['A'].double().t() @ it['B'].double()
        if seg2:
            r = r + it['A2'].double().t() @ it['B2'].double()
        ref.append((r, it['A'].double().sum(0)))
    err = {}
    for name, x3 in (('fp32', False), ('bf16x6', True)):
        res = hip.gemm_tn_grouped(its, x3=x3)
        err[name] = (max(float((c.double() - r).abs().max() / r.abs().max()) for (c, _), (r, _) in zip(res, ref)),
                     max(float((s_.double() - rs).abs().max() / rs.abs().max()) for (_, s_), (_, rs) in zip(res, ref)))
    assert err['bf16x6'][0] <= max(1.5 * err['fp32'][0], 5e-7) and err['bf16x6'][0] < 2e-6, err
    assert err['bf16x6'][1] < 2e-6, err
    with hip.wgrad_precision('bf16x6'):   # the process-wide switch routes the same call
        res2 = hip.gemm_tn_grouped(its)
    res = hip.gemm_tn_grouped(its, x3=True)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res, res2)) and hip.WGRAD_X3 is False
    # fewer partial products (psn_gemm_tn_x3_set_products): 3 = two pieces per operand (~16 significant bits), 1 = plain bf16
    # operands; the column sums stay fp32 sums of the fp32 rows in every mode; the setting is restored on exit
    for mode, lo, hi in (('bf16x3', 2e-6, 2e-5), ('bf16', 3e-4, 8e-3)):
        with hip.wgrad_precision(mode):
            resm = hip.gemm_tn_grouped(its)
        e = max(float((c.double() - r).abs().max() / r.abs().max()) for (c, _), (r, _) in zip(resm, ref))
        es = max(float((s_.double() - rs).abs().max() / rs.abs().max()) for (_, s_), (_, rs) in zip(resm, ref))
        assert lo < e < hi and es < 2e-6, (mode, e, es)
    res3 = hip.gemm_tn_grouped(its, x3=True)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res, res3))


def test_wgrad_bf16x6_passes_the_parameter_gradient_gate(cuda):
    """The split-bf16 weight-gradient kernel inside the stage-2 step (conf train.wgrad_bf16x6) and the stage-1 step (cfg
    training.wgrad_bf16x6): every parameter gradient within the 1e-3 bound of the exact path's parity gate against the oracle
    (stage 2, N = 3000 x V = 8 supervised rows) and within 2e-5 of the exact-fp32 HIP step (stage 1: one full train step from the same
    weights, draws injected) -- and the switch really changes which kernel runs (some gradient differs in its last bits)."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd import hip
    from oracle import stage2 as o2
    from tests.helpers import assert_close
    from tests.test_stage2_gpu import _run
    sd = stage2_state_dict(o2.bear_conf(), seed=5)   # (the weights of the exact path's gate, test_train_vis_bf16x6_passes_...)
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    N, L, V = 3000, 5, 8
    inp, gt = stage2_inputs(N, L, V, seed=N)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    _, _, o_g = _run(onet, o2.MainLoss, o2.NormalLoss, inp, gt, 2, nz, 'cpu')
    grads = {}
    for mode in ('fp32', 'bf16x6', 'bf16x3'):
        net = s2.PSNetwork(s2.bear_conf())
        net.load_state_dict(sd)
        net.to(cuda)
        with hip.wgrad_precision(mode):
            grads[mode] = _run(net, s2.MainLoss, s2.NormalLoss, inp, gt, 2, nz, cuda)[2]
    for k in o_g:
        assert_close(grads['bf16x6'][k].cpu(), o_g[k], 1e-3, 'grad ' + k)
        assert_close(grads['bf16x3'][k].cpu(), o_g[k], 1e-3, 'grad (three products) ' + k)
    vis = [k for k in o_g if k.startswith('visibility_net') and k.endswith('weight') and tuple(o_g[k].shape) == (256, 256)]
    assert vis and any(not torch.equal(grads['bf16x6'][k], grads['fp32'][k]) for k in vis)
    # stage 1: one train step, exact vs split weight gradients
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.synthetic import stage1_batch
    from tests.helpers import stage1_cfg, stage1_state_dict
    res = {}
    for flag in (False, True):
        cfg = stage1_cfg('bunny', **{'training.n_training_points': 256, 'training.wgrad_bf16x6': flag})
        net = NeuralNetwork(cfg)
        net.load_state_dict(stage1_state_dict(cfg, seed=21))
        tr = Trainer(Renderer(net, cfg, device=cuda), torch.optim.SGD(net.parameters(), lr=0.0), cfg, device=cuda)
        assert tr.wgrad_bf16x6 is flag
        batch = {k: v.to(cuda) for k, v in stage1_batch(cfg, h=48, w=64, seed=4).items()}
        gen = torch.Generator().manual_seed(5)
        pix = torch.stack([torch.randint(0, 64, (256,), generator=gen).float(), torch.randint(0, 48, (256,), generator=gen).float()], -1)[None]
        noise = {'full': torch.rand(256, 64, generator=gen).to(cuda), 'nbr_full': torch.rand(256, 3, generator=gen).to(cuda)}
        terms = tr.train_step(batch, it=1500, pix=pix, noise=noise)
        res[flag] = (float(terms['loss'].detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert res[True][0] == res[False][0]
    worst = max(float((res[True][1][k] - res[False][1][k]).abs().max() / res[False][1][k].abs().max().clamp_min(1e-20)) for k in res[False][1])
    assert 0.0 < worst <= 2e-5, worst


def test_chain_bf16x3_geometry_field_vs_exact_and_float64(cuda):
    """The split-bf16 form of the chain engine (PsnMlpDesc.w_format = PSN_W_BF16X2, ops.chain_precision('bf16x3'): the matrix work of
    stage1/model/network.py:85-120 -- value pass, gradient sweep and their two adjoints -- as three bf16 partial products on
    v_mfma_f32_16x16x32_bf16, activation programs / dumps / epilogues in fp32) on the geometric-init BEAR network: outputs, the
    field gradient and every parameter gradient of a double-backward objective against a float64 evaluation by torch autograd;
    the exact chains on the same inputs are the yardstick.  Bounds: values 1e-4, gradients 1e-3 of the tensor's largest entry
    (the parameter-gradient bound of the parity gate)."""
    from psnerf_amd import fused, ops
    from psnerf_amd.stage1 import NeuralNetwork
    from tests.helpers import stage1_cfg, stage1_state_dict
    cfg = stage1_cfg('bear')
    net = NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=21))
    net.to(cuda)
    params = [p.detach().clone().requires_grad_() for p in net._geo_params()]
    P64 = [p.detach().double().requires_grad_() for p in params]
    Q = 3000
    pts = ((torch.rand(Q, 3, generator=torch.Generator().manual_seed(3)) - 0.5) * 1.6).to(cuda)
    octaves, skips, scale = net.octaves_pe, tuple(net.skips), 1.0 / net.rescale

    def objective(logit, feat, grad):
        return torch.sigmoid(logit * -10.0).sum() + feat.sum() * 0.1 + (grad * grad).sum()

    def run(mode):
        for p in params:
            p.grad = None
        with ops.chain_precision(mode):
            chains = fused.pack_geo_chains(params[0::2], params[1::2], list(skips), net.d_pe, single_dump=ops.GEO_SINGLE_DUMP, x3=ops.CHAIN_X3)
            logit, feat, grad = ops.GeoFieldFused.apply(pts, octaves, scale, skips, True, chains, None, *params)
            objective(logit, feat, grad).backward()
        return [logit.detach(), feat.detach(), grad.detach()] + [p.grad.clone() for p in params]

    x = pts.double().requires_grad_()
    xs = x * scale
    pe = torch.cat([xs] + [f(xs * 2.0 ** k) for k in range(octaves) for f in (torch.sin, torch.cos)], -1)
    h = pe
    n = len(P64) // 2
    for l in range(n):
        if l in skips:
            h = torch.cat([h, pe], -1)  # (effective weights: the 1 / sqrt(2) of network.py:90-91 is folded in)
        h = h @ P64[2 * l].t() + P64[2 * l + 1]
        if l < n - 1:
            h = torch.nn.functional.softplus(h, beta=100)
    g64 = torch.autograd.grad(h[:, :1].sum(), x, create_graph=True)[0]
    objective(h[:, :1], h[:, 1:], g64).backward()
    ref = [h[:, :1].detach(), h[:, 1:].detach(), g64.detach()] + [p.grad for p in P64]
    err = {}
    for mode in ('fp32', 'bf16x3'):
        got = run(mode)
        err[mode] = [float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(got, ref)]
    print('values (logit, feat, grad) fp32 %s  bf16x3 %s' % (err['fp32'][:3], err['bf16x3'][:3]))
    print('worst parameter gradient fp32 %.2e  bf16x3 %.2e' % (max(err['fp32'][3:]), max(err['bf16x3'][3:])))
    assert max(err['fp32'][:2]) < 2e-5 and max(err['fp32'][2:]) < 1e-4, err['fp32']     # (the reference point: exact chains)
    assert max(err['bf16x3'][:2]) < 1e-4, err['bf16x3'][:3]
    assert max(err['bf16x3'][2:]) < 1e-3, err['bf16x3']
    assert max(err['bf16x3'][3:]) > max(err['fp32'][3:])   # (the split form really ran)
    from psnerf_amd import hip
    pk = fused.pack_geo_chains(params[0::2], params[1::2], list(skips), net.d_pe, x3=True)['fwd']
    assert pk.desc.w_format == hip.W_BF16X2


def test_chain_bf16x3_train_steps_vs_exact(cuda):
    """cfg training.chain_precision = 'bf16x3' (stage 1) and conf train.chain_precision = 'bf16x3' (stage 2; the V-row and
    normal / albedo backward chains): loss and every parameter gradient of one train step against the exact step on the same
    inputs and draws, within the parity gate's parameter-gradient bound (1e-3 of the tensor's largest entry)."""
    from psnerf_amd import stage2 as s2
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.synthetic import stage1_batch
    from tests.helpers import stage1_cfg, stage1_state_dict
    from tests.test_stage2_gpu import _run
    res = {}
    for mode in (None, 'bf16x3'):
        cfg = stage1_cfg('bunny', **{'training.n_training_points': 256, 'training.chain_precision': mode})
        net = NeuralNetwork(cfg)
        net.load_state_dict(stage1_state_dict(cfg, seed=21))
        tr = Trainer(Renderer(net, cfg, device=cuda), torch.optim.SGD(net.parameters(), lr=0.0), cfg, device=cuda)
        assert tr.chain_mode == mode
        batch = {k: v.to(cuda) for k, v in stage1_batch(cfg, h=48, w=64, seed=4).items()}
        gen = torch.Generator().manual_seed(5)
        pix = torch.stack([torch.randint(0, 64, (256,), generator=gen).float(), torch.randint(0, 48, (256,), generator=gen).float()], -1)[None]
        noise = {'full': torch.rand(256, 64, generator=gen).to(cuda), 'nbr_full': torch.rand(256, 3, generator=gen).to(cuda)}
        terms = tr.train_step(batch, it=1500, pix=pix, noise=noise)
        res[mode] = (float(terms['loss'].detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert abs(res['bf16x3'][0] - res[None][0]) <= 1e-4 * abs(res[None][0]), (res['bf16x3'][0], res[None][0])
    worst = max(float((res['bf16x3'][1][k] - res[None][1][k]).abs().max() / res[None][1][k].abs().max().clamp_min(1e-20)) for k in res[None][1])
    print('stage 1: loss %.6f / %.6f, worst parameter gradient difference %.2e' % (res['bf16x3'][0], res[None][0], worst))
    assert 0.0 < worst <= 1e-3, worst
    # stage 2
    sd = stage2_state_dict(s2.bear_conf(), seed=5)
    N, L, V = 3000, 5, 8
    inp, gt = stage2_inputs(N, L, V, seed=N)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    from psnerf_amd import ops
    grads = {}
    for mode in ('fp32', 'bf16x3'):
        net2 = s2.PSNetwork(s2.bear_conf())
        net2.load_state_dict(sd)
        net2.to(cuda)
        with ops.chain_precision(mode):
            grads[mode] = _run(net2, s2.MainLoss, s2.NormalLoss, inp, gt, 2, nz, cuda)[2]
    worst2 = max(float((grads['bf16x3'][k] - grads['fp32'][k]).abs().max() / grads['fp32'][k].abs().max().clamp_min(1e-20)) for k in grads['fp32'])
    print('stage 2: worst parameter gradient difference %.2e' % worst2)
    assert 0.0 < worst2 <= 1e-3, worst2


def test_bf16x3_train_steps_vs_oracle(cuda):
    """The contract of the split-bf16 path: bf16x3 = two bf16 pieces per operand, three partial products, ~1e-5 relative
    per multiply.  Against the ORACLE (not the exact HIP step): one stage-2 train step with every large MFMA
    stream on it (conf train.vis_bf16x3 + train.chain_precision + train.wgrad_precision) keeps the exact path's gates -- loss
    terms 1e-4, parameter and light-table gradients 1e-3 max-normalised; one stage-1 train step (training.chain_precision +
    training.wgrad_precision; march, root finder, forward of the appearance net exact) keeps loss terms at the exact path's
    bounds and the parameters after the Adam step at the same bound as test_train_step_vs_oracle.  What bf16x3 does NOT keep is
    north_star's 1e-4 on stage-1 EVALUATION renders through the x3 occupancy engine (2e-3 on rgb, <= 2 mask flips: the test below)."""
    from oracle import stage1 as o1, stage2 as o2
    from psnerf_amd import ops, stage2 as s2
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    from psnerf_amd.synthetic import stage1_batch
    from tests.helpers import assert_close, stage1_cfg, stage1_state_dict
    # ---- stage 2
    over = {'train.vis_bf16x3': True, 'train.chain_precision': 'bf16x3', 'train.wgrad_precision': 'bf16x3'}
    sd = stage2_state_dict(o2.bear_conf(), seed=5)
    N, L, V, NL = 3000, 6, 8, 24
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    light_init[:, 2] = light_init[:, 2].abs() + 0.2
    onet = o2.PSNetwork(o2.bear_conf())
    onet.load_state_dict(sd)
    ostep = o2.TrainStep(onet, o2.bear_conf(), NL, light_init)
    net = s2.PSNetwork(s2.bear_conf(**over))
    net.load_state_dict(sd)
    net.to(cuda)
    step = s2.TrainStep(net, s2.bear_conf(**over), NL, light_init.to(cuda), cuda)
    ostep.cur_iter = step.cur_iter = 5001
    inp, gt = stage2_inputs(N, L, V, seed=77, with_surface_idx=True)
    ns = int(inp['surface_mask'].sum())
    nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(1)) * 0.01
    l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(2))[:L]
    ot, _ = ostep.step({k: v for k, v in inp.items() if k != 'surface_idx'}, gt, l_slt, train_order=False, noise={'xyz': nz})
    ops.reset_hits()
    with ops.strict():
        pt, _, _, _ = step._fwd_bwd({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt.to(cuda),
                                    noise={'xyz': nz.to(cuda)})
    for k in ('total', 'sg_rgb_loss', 'vis_loss', 'normal_loss', 'albedo_smooth_loss', 'rough_smooth_loss'):
        assert_close(float(pt[k].detach()), float(ot[k].detach()), 1e-4, 'stage 2 bf16x3 ' + k, atol=0.0)
    gr = {k: p.grad.detach() for k, p in net.named_parameters() if p.grad is not None}
    gr['__light_dir'], gr['__light_int'] = step.light_para.weight.grad.detach(), step.light_inten_para.weight.grad.detach()
    ogr = {k: p.grad for k, p in onet.named_parameters() if p.grad is not None}
    ogr['__light_dir'], ogr['__light_int'] = ostep.light_para.weight.grad.to_dense(), ostep.light_inten_para.weight.grad.to_dense()
    assert sorted(gr) == sorted(ogr)
    for k in sorted(gr):
        assert_close(gr[k].cpu(), ogr[k], 1e-3, 'stage 2 bf16x3 grad ' + k)
    # ---- stage 1 (the protocol of tests/test_stage1_gpu.py::test_train_step_vs_oracle)
    cfg = stage1_cfg('bunny', **{'training.n_training_points': 128})
    cfg_x = stage1_cfg('bunny', **{'training.n_training_points': 128, 'training.chain_precision': 'bf16x3', 'training.wgrad_precision': 'bf16x3'})
    sd1 = stage1_state_dict(cfg, seed=21)
    onet1 = o1.NeuralNetwork(cfg)
    onet1.load_state_dict(sd1)
    otr = o1.Trainer(o1.Renderer(onet1, cfg), torch.optim.Adam(onet1.parameters(), lr=1e-4), cfg)
    net1 = NeuralNetwork(cfg_x)
    net1.load_state_dict(sd1)
    tr = Trainer(Renderer(net1, cfg_x, device=cuda), torch.optim.Adam(net1.parameters(), lr=1e-4), cfg_x, device=cuda)
    assert tr.chain_mode == 'bf16x3'
    batch = stage1_batch(cfg, h=48, w=64, seed=4)
    for it in (1000, 1001):
        gen = torch.Generator().manual_seed(it)
        pix = torch.stack([torch.randint(0, 64, (128,), generator=gen).float(), torch.randint(0, 48, (128,), generator=gen).float()], -1)[None]
        with torch.no_grad():
            dry = o1.Renderer(onet1, cfg)(pix, batch['img.camera_mat'], batch['img.world_mat'], batch['img.scale_mat'], 'unisurf',
                                          add_noise=False, eval_=True, it=it)
        n_hit = int(dry['mask_pred'].sum())
        noise = {'miss': torch.rand(1, 128 - n_hit, 64, generator=gen), 'hit': torch.rand(1, n_hit, 64, generator=gen), 'nbr': torch.rand(n_hit, 3, generator=gen)}
        o_t = otr.train_step(batch, it=it, pix=pix, noise=noise)
        p_t = tr.train_step(batch, it=it, pix=pix, noise={k: v.to(cuda) for k, v in noise.items()})
        for k in o_t:
            assert_close(float(p_t[k].detach()), float(o_t[k].detach()), 1e-3 if k == 'grad_loss' else 2e-4, 'stage 1 bf16x3 %s it%d' % (k, it), atol=0.0)
    osd = onet1.state_dict()
    for k, v in net1.state_dict().items():
        d = (v.cpu() - osd[k]).abs()
        assert float(d.max()) <= 2 * 2 * 1e-4 + 1e-6 and float(d.mean()) <= 1e-5, 'param %s max diff %.3e mean %.3e' % (k, float(d.max()), float(d.mean()))


def test_occupancy_and_march_sweep_on_split_bf16_weight_stages(cuda):
    """NeuralNetwork.inference_precision = 'bf16x3': the gradient-free occupancy queries and the one-launch ray-march sweep through
    the exact engine's kernels on split-bf16 weight stages (PSN_W_BF16X2).  Occupancy within 2e-4 of the fp32 engine's (the sigmoid
    of -10 x logit amplifies the logit's 1e-5), and the march -- first crossing + secant refinement on the exact root finder --
    agrees with the exact march: same hit mask except for grazing rays, depths within 1e-4."""
    from psnerf_amd.stage1 import NeuralNetwork, Renderer
    from psnerf_amd.synthetic import stage1_batch
    from tests.helpers import stage1_cfg, stage1_state_dict
    cfg = stage1_cfg('bear')
    net = NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=21))
    net.to(cuda).eval()
    pts = ((torch.rand(20000, 3, generator=torch.Generator().manual_seed(2)) - 0.5) * 1.8).to(cuda)
    with torch.no_grad():
        occ = {}
        for prec in ('fp32', 'bf16x3'):
            net.inference_precision = prec
            occ[prec] = net.occupancy(pts).clone()
    d = float((occ['bf16x3'] - occ['fp32']).abs().max())
    assert 0.0 < d < 2e-4, d
    ren = Renderer(net, cfg, device=cuda)
    batch = {k: v.to(cuda) for k, v in stage1_batch(cfg, h=48, w=64, seed=4).items()}
    gen = torch.Generator().manual_seed(5)
    pix = torch.stack([torch.randint(0, 64, (1024,), generator=gen).float(), torch.randint(0, 48, (1024,), generator=gen).float()], -1)[None].to(cuda)
    res = {}
    with torch.no_grad():  # (the evaluation render of a pixel set: march + secant + volume rendering, gradient-free)
        for prec in ('fp32', 'bf16x3'):
            net.inference_precision = prec
            out = ren(pix, batch['img.camera_mat'], batch['img.world_mat'], batch['img.scale_mat'], 'unisurf', add_noise=False, eval_=True, it=6000)
            res[prec] = (out['mask_pred'].clone(), out['rgb'].clone())
    net.inference_precision = 'fp32'
    flips = int((res['fp32'][0] != res['bf16x3'][0]).sum())
    assert flips <= 2, flips
    same = (res['fp32'][0] == res['bf16x3'][0]).reshape(-1)
    drgb = float((res['fp32'][1] - res['bf16x3'][1]).reshape(-1, 3)[same].abs().max())
    assert drgb < 2e-3, drgb
