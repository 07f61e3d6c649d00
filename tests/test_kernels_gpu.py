"""HIP kernels (through the C ABI) against the CPU oracle / golden vectors.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, assert_close, stage1_state_dict, stage2_state_dict, stage1_cfg

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize('S', [64, 96, 128])
def test_composite_golden(cuda, S):
    from psnerf_amd import hip
    g = np.load(os.path.join(GOLDEN, 'stage1_composite.npz'))
    alpha, rgb = T(g['alpha%d' % S], cuda), T(g['rgb%d' % S], cuda)
    w, out, acc = hip.composite_fwd(alpha, rgb, True)
    assert_close(w.cpu(), g['w%d' % S], 2e-6, 'w')
    assert_close(out.cpu(), g['out%d' % S], 2e-6, 'rgb')
    assert_close(acc.cpu(), g['acc%d' % S], 2e-6, 'acc')
    w0, out0, acc0 = hip.composite_fwd(alpha, rgb, True, need_weights=False)  # the training path (flat-layout kernel for 64 < S <= 128)
    assert w0 is None
    assert_close(out0.cpu(), g['out%d' % S], 2e-6, 'rgb (no weights)')
    assert_close(acc0.cpu(), g['acc%d' % S], 2e-6, 'acc (no weights)')
    da, dc = hip.composite_bwd(alpha, rgb, T(g['c1_%d' % S], cuda), T(g['c2_%d' % S], cuda), True)
    assert_close(dc.cpu(), g['drgb%d' % S], 2e-6, 'd_rgb')
    assert_close(da.cpu(), g['dalpha%d' % S], 1e-4, 'd_alpha')  # /t with t ~ 1e-6 on the alpha==1 rows


@pytest.mark.parametrize('N,S', [(1, 1), (5, 7), (1000, 63), (333, 65), (4096, 128), (17, 300), (3, 1024),
                                 (21, 100), (257, 96), (6, 127), (50, 256), (9, 512), (4, 1000), (70001, 2)])  # vector / scalar row paths, idle lanes
def test_composite_shapes(cuda, N, S):
    from psnerf_amd import hip
    from oracle import stage1 as o1
    g = torch.Generator().manual_seed(N * 1000 + S)
    alpha = torch.rand(N, S, generator=g, dtype=torch.float64) * 0.3
    rgb = torch.rand(N, S, 3, generator=g, dtype=torch.float64)
    a64, c64 = alpha.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    w_ref, rgb_ref = o1.alpha_composite(a64, c64)
    acc_ref = w_ref.sum(-1)
    c1, c2 = torch.randn(N, 3, generator=g, dtype=torch.float64), torch.randn(N, generator=g, dtype=torch.float64)
    ((rgb_ref * c1).sum() + (acc_ref * c2).sum()).backward()
    a32, c32 = alpha.float().to(cuda), rgb.float().to(cuda)
    w, out, acc = hip.composite_fwd(a32, c32, False)
    assert_close(w.cpu(), w_ref.detach(), 5e-6, 'w')
    assert_close(out.cpu(), rgb_ref.detach(), 5e-6, 'rgb')
    for white in (False, True):  # colours without the weights output
        w0, out0, acc0 = hip.composite_fwd(a32, c32, white, need_weights=False)
        assert w0 is None
        assert_close(out0.cpu(), rgb_ref.detach() + ((1 - acc_ref.detach())[:, None] if white else 0), 5e-6, 'rgb (no weights)')
        assert_close(acc0.cpu(), acc_ref.detach(), 5e-6, 'acc (no weights)')
    da, dc = hip.composite_bwd(a32, c32, c1.float().to(cuda), c2.float().to(cuda), False)
    assert_close(dc.cpu(), c64.grad, 5e-6, 'd_rgb')
    assert_close(da.cpu(), a64.grad, 2e-5, 'd_alpha')
    # acc-only form (light visibility)
    w2, none_, acc2 = hip.composite_fwd(a32, None, False, need_weights=False)
    assert w2 is None and none_ is None
    assert_close(acc2.cpu(), acc_ref.detach(), 5e-6, 'acc only')


def test_composite_known_answers(cuda):
    from psnerf_amd import hip
    z = torch.zeros(9, 80, device=cuda)
    w, rgb, acc = hip.composite_fwd(z, torch.rand(9, 80, 3, device=cuda), True)
    assert float(acc.abs().max()) == 0 and float((rgb - 1).abs().max()) == 0  # white background
    o = torch.ones(9, 80, device=cuda)
    w, rgb, acc = hip.composite_fwd(o, torch.rand(9, 80, 3, device=cuda), False)
    assert float((w[:, 0] - 1).abs().max()) == 0 and float(w[:, 1:].abs().max()) < 2e-6


@pytest.mark.parametrize('n_freqs,stride', [(6, 39), (6, 64), (10, 64), (4, 27), (0, 3)])
def test_pe(cuda, n_freqs, stride):
    from psnerf_amd import hip
    from oracle import stage1 as o1
    from oracle import stage2 as o2
    g = torch.Generator().manual_seed(n_freqs)
    x = (torch.rand(1001, 3, generator=g) * 2 - 1)
    ref = o1.positional_encoding(x, n_freqs)
    if n_freqs > 0:
        assert torch.equal(ref, o2.embed(x, n_freqs))  # both reference encodings coincide
    out = hip.pe_encode(x.to(cuda), n_freqs, stride).cpu()
    assert_close(out[:, :ref.shape[1]], ref, 1e-6, 'pe')
    assert float(out[:, ref.shape[1]:].abs().max() if stride > ref.shape[1] else 0.0) == 0
    xg = x.double().requires_grad_(True)
    d_out = torch.randn(1001, stride, generator=g)
    (o1.positional_encoding(xg, n_freqs) * d_out[:, :ref.shape[1]].double()).sum().backward()
    dx = hip.pe_encode_bwd(x.to(cuda), d_out.to(cuda), n_freqs).cpu()
    assert_close(dx, xg.grad, 1e-5, 'pe bwd')
    # the gradient given as two pieces that are column ranges of wider tensors (summed inside the kernel)
    w = ref.shape[1]
    wide1, wide2 = torch.randn(1001, 256, generator=g), torch.randn(1001, 256, generator=g)
    xg2 = x.double().requires_grad_(True)
    (o1.positional_encoding(xg2, n_freqs) * (wide1[:, :w] + wide2[:, 200 - w:200]).double()).sum().backward()
    dx2 = hip.pe_encode_bwd(x.to(cuda), wide1.to(cuda)[:, :w], n_freqs, add=wide2.to(cuda)[:, 200 - w:200]).cpu()
    assert_close(dx2, xg2.grad, 1e-5, 'pe bwd, two strided pieces')


@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (1, 1, 1), (257, 129, 33), (1000, 256, 126), (300, 217, 256),
                                   (513, 3, 128), (64, 289, 300)])
def test_gemm_layouts(cuda, ta, tb, M, N, K):
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = (A.double() @ B.double())
    # asymmetric operands by construction (random) -> catches transposed C writes
    Ad = (A.t().contiguous() if ta else A).to(cuda)
    Bd = (B.t().contiguous() if tb else B).to(cuda)
    out = hip.gemm(Ad, Bd, trans_a=ta, trans_b=tb).cpu()
    assert_close(out, ref, 2e-6 * max(1, K ** 0.5), 'gemm')


def test_gemm_epilogues_and_strides(cuda):
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(5)
    M, N, K = 700, 256, 126
    buf = torch.randn(M, 384, generator=g).to(cuda)        # A is a column slice of a wider buffer
    A = buf[:, 256:256 + K]
    W = torch.randn(N, K, generator=g).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    z = (A.double().cpu() @ W.double().cpu().t()) + b.double().cpu()
    assert_close(hip.gemm(A, W, trans_b=True, bias=b, epi=hip.EPI_BIAS).cpu(), z, 5e-6, 'bias')
    assert_close(hip.gemm(A, W, trans_b=True, bias=b, epi=hip.EPI_BIAS_RELU).cpu(), z.clamp(min=0), 5e-6, 'relu')
    assert_close(hip.gemm(A, W, trans_b=True, bias=b, epi=hip.EPI_BIAS_SIGMOID).cpu(), torch.sigmoid(z), 5e-6, 'sigmoid')
    zs = z * 0.02
    s_out = torch.empty(M, N, device=cuda)
    sp = hip.gemm(A, W * 0.02, trans_b=True, bias=b * 0.02, epi=hip.EPI_BIAS_SOFTPLUS, aux_out=s_out).cpu()
    assert_close(sp, torch.nn.functional.softplus(zs, beta=100, threshold=20), 1e-5, 'softplus')
    assert_close(s_out.cpu(), torch.sigmoid(100 * zs), 2e-4, 'softplus aux (sigmoid amplifies 100x)')
    aux = torch.randn(M, N, generator=g).to(cuda)
    raw = A.double().cpu() @ W.double().cpu().t()
    assert_close(hip.gemm(A, W, trans_b=True, epi=hip.EPI_MUL_AUX, aux_in=aux).cpu(), raw * aux.double().cpu(), 5e-6, 'mul')
    assert_close(hip.gemm(A, W, trans_b=True, epi=hip.EPI_MUL_POS, aux_in=aux).cpu(), raw * (aux.cpu() > 0), 5e-6, 'pos')
    # output into a column slice, accumulate
    wide = torch.zeros(M, 512, device=cuda)
    hip.gemm(A, W, trans_b=True, out=wide[:, 100:100 + N])
    hip.gemm(A, W, trans_b=True, out=wide[:, 100:100 + N], epi=hip.EPI_ACCUM)
    assert_close(wide[:, 100:100 + N].cpu(), 2 * raw, 5e-6, 'accum')
    assert float(wide[:, :100].abs().max()) == 0 and float(wide[:, 100 + N:].abs().max()) == 0


@pytest.mark.parametrize('split', [1, 3, 16, 64])
def test_gemm_splitk_weight_grad(cuda, split):
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(split)
    Q, O, I = 5000, 256, 382
    dZ = torch.randn(Q, O, generator=g)
    X = torch.randn(Q, I, generator=g)
    ref = dZ.double().t() @ X.double()
    out = hip.gemm(dZ.to(cuda), X.to(cuda), trans_a=True, trans_b=False, split_k=split).cpu()
    assert_close(out, ref, 1e-5, 'dW')
    cs = hip.colsum(dZ.to(cuda)).cpu()
    assert_close(cs, dZ.double().sum(0), 1e-5, 'colsum')
    # bias gradient as a by-product of the weight-gradient GEMM (column sums of A), ragged M and strided A
    for cols in (256, 217, 1):
        cs2 = torch.empty(cols, device=cuda)
        a_view = dZ.to(cuda)[:, :cols]
        out3 = hip.gemm(a_view, X.to(cuda), trans_a=True, split_k=split, colsum_a=cs2).cpu()
        assert_close(out3, dZ[:, :cols].double().t() @ X.double(), 1e-5, 'dW with colsum (%d)' % cols)
        assert_close(cs2.cpu(), dZ[:, :cols].double().sum(0), 1e-5, 'colsum by-product (%d)' % cols)
    if split > 1:
        # vectorised reduce (N % 4 == 0) with accumulation into a strided view, and the scalar fallback (N = 217)
        base = torch.randn(O, 380 + 8, generator=g)
        acc = base.clone().to(cuda)
        hip.gemm(dZ.to(cuda), X[:, :380].contiguous().to(cuda), trans_a=True, split_k=split, out=acc[:, 4:384], epi=hip.EPI_ACCUM)
        assert_close(acc[:, 4:384].cpu(), base[:, 4:384].double() + dZ.double().t() @ X[:, :380].double(), 1e-5, 'dW accumulate')
        assert torch.equal(acc[:, :4].cpu(), base[:, :4]) and torch.equal(acc[:, 384:].cpu(), base[:, 384:])
        out2 = hip.gemm(dZ.to(cuda), X[:, :217].contiguous().to(cuda), trans_a=True, split_k=split).cpu()
        assert_close(out2, dZ.double().t() @ X[:, :217].double(), 1e-5, 'dW (N = 217)')


@pytest.mark.parametrize('Q,split', [(5000, 4), (1234, 1), (40000, 16)])
def test_gemm_tn_grouped(cuda, Q, split):
    """All weight gradients of a backward pass in one launch: ragged shapes, strided views, two-product items
    (value pass + gradient sweep of a shared layer), accumulation, bias-gradient by-products."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(Q)
    dZ = [torch.randn(Q, 256, generator=g) for _ in range(4)]
    X = [torch.randn(Q, 256, generator=g) for _ in range(4)]
    pe = torch.randn(Q, 64, generator=g)
    base = torch.randn(256, 256, generator=g)
    d = lambda t: t.to(cuda)
    dZd, Xd, ped = [d(t) for t in dZ], [d(t) for t in X], d(pe)
    acc = d(base.clone())
    items = [
        dict(A=dZd[0], B=Xd[0], colsum=True),
        dict(A=dZd[1][:, :217], B=Xd[1], colsum=True),                      # ragged M, strided A
        dict(A=dZd[2], B=ped[:, :39], colsum=True),                         # N = 39, strided B
        dict(A=dZd[3], B=Xd[3], A2=dZd[0], B2=Xd[1], colsum=True),          # two products into one gradient
        dict(A=dZd[1], B=Xd[2], out=acc, accumulate=True),                  # accumulate into an existing tensor
        dict(A=dZd[2][:, :1], B=Xd[0]),                                     # M = 1
    ]
    res = hip.gemm_tn_grouped(items, split)
    D = lambda t: t.double()
    refs = [D(dZ[0]).t() @ D(X[0]), D(dZ[1][:, :217]).t() @ D(X[1]), D(dZ[2]).t() @ D(pe[:, :39]),
            D(dZ[3]).t() @ D(X[3]) + D(dZ[0]).t() @ D(X[1]), D(base) + D(dZ[1]).t() @ D(X[2]), D(dZ[2][:, :1]).t() @ D(X[0])]
    cs_refs = [D(dZ[0]).sum(0), D(dZ[1][:, :217]).sum(0), D(dZ[2]).sum(0), D(dZ[3]).sum(0), None, None]
    for i, ((C, cs), ref, cref) in enumerate(zip(res, refs, cs_refs)):
        assert_close(C.cpu(), ref, 1e-5, 'grouped dW %d' % i)
        if cref is None:
            assert cs is None
        else:
            assert_close(cs.cpu(), cref, 1e-5, 'grouped colsum %d' % i)
    # same results as the one-at-a-time path, and deterministic
    one = hip.gemm(dZd[0], Xd[0], trans_a=True, split_k=split)
    assert_close(res[0][0].cpu(), one.cpu(), 1e-5, 'grouped vs single')  # different K slicing: rounding only
    res2 = hip.gemm_tn_grouped(items, split)  # the same grouping again: bit-identical (fixed summation order)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res[:4], res2[:4]))


def test_gemm_tn_grouped_table_operand(cuda):
    """B given as a TABLE indexed (k // b_div) % b_mod: the weight gradient of an input block that repeats per light /
    per point (rows k = v * Ns + n of the visibility supervision set) without expanding it."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(3)
    Ns, V = 777, 5
    Q = Ns * V
    dz = torch.randn(Q, 256, generator=g)
    pe_x, pe_l = torch.randn(Ns, 64, generator=g), torch.randn(V, 64, generator=g)
    d = lambda t: t.to(cuda)
    res = hip.gemm_tn_grouped([dict(A=d(dz), B=d(pe_x), b_div=1, b_mod=Ns, colsum=True),
                               dict(A=d(dz), B=d(pe_l), b_div=Ns, b_mod=V),
                               dict(A=d(dz), B=d(dz))])
    D = lambda t: t.double()
    assert_close(res[0][0].cpu(), D(dz).t() @ D(pe_x).repeat(V, 1), 1e-5, 'table operand (k % Ns)')
    assert_close(res[1][0].cpu(), D(dz).t() @ D(pe_l).repeat_interleave(Ns, dim=0), 1e-5, 'table operand (k // Ns)')
    assert_close(res[0][1].cpu(), D(dz).sum(0), 1e-5, 'colsum')
    assert_close(res[2][0].cpu(), D(dz).t() @ D(dz), 1e-5, 'plain item in the same group')
    # both tables side by side in one 128-column product
    both = hip.gemm_tn_grouped([dict(A=d(dz), B=d(pe_x), b_div=1, b_mod=Ns, B_tab2=d(pe_l), b2_div=Ns, b2_mod=V, colsum=True)])[0]
    assert both[0].shape == (256, 128)
    assert_close(both[0][:, :64].cpu(), D(dz).t() @ D(pe_x).repeat(V, 1), 1e-5, 'two tables: first')
    assert_close(both[0][:, 64:].cpu(), D(dz).t() @ D(pe_l).repeat_interleave(Ns, dim=0), 1e-5, 'two tables: second')
    assert_close(both[1].cpu(), D(dz).sum(0), 1e-5, 'two tables: colsum')


@pytest.mark.parametrize('Q', [7, 100, 4099])
def test_gemm_tn_grouped_tile256_edges(cuda, Q):
    """The one-256x256-tile-per-workgroup path (128 < M, N <= 256) on awkward shapes: K smaller than / not a multiple
    of the 16-row k-tile, operands that are column slices of narrow buffers (row stride < 256), two products."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(Q)
    A = torch.randn(Q, 132, generator=g)   # M = 130 of a 132-wide buffer
    B = torch.randn(Q, 200, generator=g)   # N = 200, row stride 200
    A2 = torch.randn(Q, 256, generator=g)
    B2 = torch.randn(Q, 204, generator=g)
    d = lambda t: t.to(cuda)
    Ad, Bd, A2d, B2d = d(A), d(B), d(A2), d(B2)
    items = [dict(A=Ad[:, :130], B=Bd, colsum=True),
             dict(A=A2d[:, :130], B=B2d[:, :200], A2=Ad[:, :130], B2=Bd, colsum=True),
             dict(A=A2d, B=B2d[:, :129])]
    res = hip.gemm_tn_grouped(items)
    D = lambda t: t.double()
    refs = [D(A[:, :130]).t() @ D(B), D(A2[:, :130]).t() @ D(B2[:, :200]) + D(A[:, :130]).t() @ D(B), D(A2).t() @ D(B2[:, :129])]
    for i, ((C, cs), ref) in enumerate(zip(res, refs)):
        assert_close(C.cpu(), ref, 1e-5, 'tile256 dW %d (Q=%d)' % (i, Q))
    assert_close(res[0][1].cpu(), D(A[:, :130]).sum(0), 1e-5, 'tile256 colsum 0')
    assert_close(res[1][1].cpu(), D(A2[:, :130]).sum(0), 1e-5, 'tile256 colsum 1')


@pytest.mark.parametrize('M,N', [(5000, 256), (70001, 256), (333, 128), (999, 64), (1234, 39), (3, 256)])
def test_colsum_weighted_and_plain(cuda, M, N):
    """psn_colsum: plain column sums and the n_w <= 4 weighted form (g^T H of a head with <= 4 outputs), on the 16-byte
    path (N % 4 == 0) and the scalar fallback, strided X, accumulation."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(M + N)
    X = torch.randn(M, N + 4, generator=g)
    W = torch.randn(M, 4, generator=g)
    Xd, Wd = X.to(cuda), W.to(cuda)
    D = lambda t: t.double()
    for xv, xr in ((Xd[:, :N], X[:, :N]), (Xd[:, 4:], X[:, 4:])):       # row stride N + 4; the second view is 16 B aligned too
        assert_close(hip.colsum(xv).cpu(), D(xr).sum(0), 1e-5, 'plain colsum')
        for nw in (1, 3, 4):
            out = hip.colsum(xv, row_weight=Wd[:, :nw])
            assert out.shape == (nw, N)
            assert_close(out.cpu(), D(W[:, :nw]).t() @ D(xr), 1e-5, 'weighted colsum n_w=%d' % nw)
        acc = torch.ones(N, device=cuda)
        hip.colsum(xv, out=acc, accumulate=True)
        assert_close(acc.cpu(), 1.0 + D(xr).sum(0), 1e-5, 'accumulating colsum')
    one = hip.colsum(Xd[:, :N], row_weight=Wd[:, 0].contiguous())      # a 1-D weight counts as one column
    assert_close(one.cpu(), (D(W[:, :1]).t() @ D(X[:, :N])), 1e-5, '1-D row weight')


@pytest.mark.parametrize('Q', [7, 100, 4099, 70001])
def test_gemm_tn_grouped_tall_tile_edges(cuda, Q):
    """The 256 x (<= 64)-tile path (128 < M <= 256, N <= 64: the input-block gradients of the stage-1 networks) on awkward
    shapes: K not a multiple of the k-tile, ragged M and N, column slices, two products, next to items of the other paths."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(Q)
    A = torch.randn(Q, 256, generator=g)
    A2 = torch.randn(Q, 132, generator=g)  # M = 130 of a 132-wide buffer
    pe = torch.randn(Q, 64, generator=g)
    x = torch.randn(Q, 36, generator=g)    # N = 33 of a 36-wide buffer
    d = lambda t: t.to(cuda)
    Ad, A2d, ped, xd = d(A), d(A2), d(pe), d(x)
    items = [dict(A=Ad, B=ped[:, :39], A2=Ad[:, :256], B2=ped[:, 4:43], colsum=True),   # two products, N = 39
             dict(A=A2d[:, :130], B=xd[:, :33], colsum=True),                          # ragged M and N
             dict(A=Ad, B=ped),                                                        # N = 64 exactly
             dict(A=Ad[:, :129], B=ped[:, :1]),                                        # N = 1
             dict(A=Ad, B=Ad, colsum=True),                                            # 256 x 256 path in the same group
             dict(A=Ad[:, :100], B=ped[:, :39])]                                       # M <= 128: the 128 x 128 tiles
    res = hip.gemm_tn_grouped(items)
    D = lambda t: t.double()
    refs = [D(A).t() @ D(pe[:, :39]) + D(A).t() @ D(pe[:, 4:43]), D(A2[:, :130]).t() @ D(x[:, :33]), D(A).t() @ D(pe),
            D(A[:, :129]).t() @ D(pe[:, :1]), D(A).t() @ D(A), D(A[:, :100]).t() @ D(pe[:, :39])]
    for i, ((C, cs), ref) in enumerate(zip(res, refs)):
        assert C.shape == ref.shape
        assert_close(C.cpu(), ref, 1e-5, 'tall dW %d (Q=%d)' % (i, Q))
    assert_close(res[0][1].cpu(), D(A).sum(0), 1e-5, 'tall colsum 0')
    assert_close(res[1][1].cpu(), D(A2[:, :130]).sum(0), 1e-5, 'tall colsum 1')
    res2 = hip.gemm_tn_grouped(items)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res, res2))  # deterministic


def test_fused_visibility_mlp(cuda):
    """mlp_infer on the stage2 visibility net == oracle MLP (stage2/model/renderer.py:191-200)."""
    from psnerf_amd import hip, fused
    from oracle import stage2 as o2
    conf = o2.bear_conf()
    net = o2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=31))
    vn = net.visibility_net
    g = torch.Generator().manual_seed(8)
    Ns, L = 333, 7
    x = torch.rand(Ns, 3, generator=g) * 1.2 - 0.6
    l = torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1)
    with torch.no_grad():
        inp = torch.cat([o2.embed(x, 10).tile(L, 1), o2.embed(l, 10).repeat_interleave(Ns, dim=0)], -1)
        ref = vn(inp)
    packed = fused.pack_relu_mlp([m.weight.to(cuda) for m in vn.linears], [m.bias.to(cuda) for m in vn.linears],
                                 63, 63, skip_at=4)
    ta = hip.pe_encode(x.to(cuda), 10, 64)
    tb = hip.pe_encode(l.to(cuda), 10, 64)
    out = packed(ta, L * Ns, a_div=1, a_mod=Ns, tab_b=tb, b_div=Ns, b_mod=L)
    assert out.shape == (L * Ns, 1)
    assert_close(out.cpu(), ref, 1e-4, 'visibility net (input block through init tables)')
    packed2 = fused.pack_relu_mlp([m.weight.to(cuda) for m in vn.linears], [m.bias.to(cuda) for m in vn.linears],
                                  63, 63, skip_at=4, precompute=False)
    out2 = packed2(ta, L * Ns, a_div=1, a_mod=Ns, tab_b=tb, b_div=Ns, b_mod=L)
    assert_close(out2.cpu(), ref, 1e-4, 'visibility net (input block as MFMA k-tiles)')


@pytest.mark.parametrize('width', [64, 128, 256])
def test_fused_relu_net_widths(cuda, width):
    """ops.FusedReluNet (lean forward with dumps + backward chain + grouped weight gradients) for both hidden widths of
    the engine against a float64 restatement of stage2/model/renderer.py:17-49: outputs and every parameter gradient,
    ragged row count, skip connection, sigmoid head."""
    from psnerf_amd import hip, ops
    g = torch.Generator().manual_seed(width)
    Q, din, skip_at = 1000, 63, 2
    dims_in = [din, width, width, width + din, width]
    dims_out = [width, width, width, width, 3]
    Ws = [torch.randn(o, i, generator=g) * (1.2 / i ** 0.5) for i, o in zip(dims_in, dims_out)]
    bs = [torch.randn(o, generator=g) * 0.05 for o in dims_out]
    x = torch.rand(Q, 3, generator=g) - 0.5
    c_out = torch.randn(Q, 3, generator=g)
    pe = hip.pe_encode(x.to(cuda), 10, 64)
    params = []
    for W, b in zip(Ws, bs):
        params += [W.to(cuda).requires_grad_(), b.to(cuda).requires_grad_()]
    out = ops.FusedReluNet.apply(pe, din, skip_at, True, width, None, *params)
    (out * c_out.to(cuda)).sum().backward()
    # float64 reference
    Wd = [W.double().requires_grad_() for W in Ws]
    bd = [b.double().requires_grad_() for b in bs]
    xin = pe.cpu().double()[:, :din]
    h = xin
    for l in range(5):
        if l - 1 == skip_at:
            h = torch.cat([h, xin], dim=1)
        h = h @ Wd[l].t() + bd[l]
        if l < 4:
            h = torch.relu(h)
    ref = torch.sigmoid(h)
    (ref * c_out.double()).sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), 1e-5, 'output')
    for l in range(5):
        assert_close(params[2 * l].grad.cpu(), Wd[l].grad, 2e-5, 'dW%d' % l)
        assert_close(params[2 * l + 1].grad.cpu(), bd[l].grad, 2e-5, 'db%d' % l)
    # the backward chain on the forward's sign-bit words (the default, PSN_ACT_RELU_BITS) and on the activation rows as masks
    # (PSN_ACT_RELU_MASK) are the same function: every gradient bit for bit
    assert ops.RELU_SIGN_BITS
    ops.RELU_SIGN_BITS = False
    try:
        p2 = [p.detach().clone().requires_grad_() for p in params]
        out2 = ops.FusedReluNet.apply(pe, din, skip_at, True, width, None, *p2)
        (out2 * c_out.to(cuda)).sum().backward()
    finally:
        ops.RELU_SIGN_BITS = True
    assert torch.equal(out2, out)
    for a, b in zip(p2, params):
        assert torch.equal(a.grad, b.grad)


def test_fused_geo_occupancy(cuda):
    """mlp_infer on the stage1 occupancy net == oracle forward(only_occupancy=True) (network.py:124-125)."""
    from psnerf_amd import hip, fused
    from oracle import stage1 as o1
    cfg = stage1_cfg('bunny')
    net = o1.NeuralNetwork(cfg)
    net.load_state_dict(stage1_state_dict(cfg, seed=11))
    g = torch.Generator().manual_seed(2)
    p = torch.rand(1000, 3, generator=g) * 2 - 1
    with torch.no_grad():
        ref = net(p, only_occupancy=True)
        logit_ref = net.infer_occ(p)[:, :1]
    ws = [getattr(net, 'lin%d' % l).weight().detach().to(cuda) for l in range(net.n_geo)]
    bs = [getattr(net, 'lin%d' % l).bias.detach().to(cuda) for l in range(net.n_geo)]
    packed = fused.pack_geo_occupancy(ws, bs, net.skips, 39)
    tab = hip.pe_encode(p.to(cuda), 6, 64)
    out = packed(tab, 1000)
    # sigmoid(-10 x) amplifies logit error 2.5x at most; compare on both scales
    assert_close(out.cpu(), ref, 1e-4, 'occupancy')
    packed.desc.out_act = hip.OUT_NONE
    logit = packed(tab, 1000)
    assert_close(logit.cpu(), logit_ref, 1e-4, 'occupancy logit')


def test_mlp_chain_activation_programs(cuda):
    """Every per-layer activation program of the chain engine (psn_mlp_infer with operands and dumps) against a
    float64 restatement: ragged row count, input-feature k-tiles, two operands, two dumps, HEAD side outputs."""
    from psnerf_amd import hip, fused
    g = torch.Generator().manual_seed(0)
    Q = 1000
    rn = lambda *sh: torch.randn(*sh, generator=g)
    x = rn(Q, 64)
    W = [rn(256, 64) * 0.1] + [rn(256, 256) * 0.06 for _ in range(6)]
    Wx2 = rn(256, 64) * 0.1
    Wf = rn(3, 256) * 0.06
    b = [rn(256) * 0.05 for _ in range(7)]
    bf = rn(3) * 0.05
    m1, m2, q2, t4, h5 = rn(Q, 256), rn(Q, 256), rn(Q, 256), rn(Q, 256) * 0.01, rn(Q, 256)
    s4 = torch.rand(Q, 256, generator=g)
    layers = [
        dict(w_in=W[0], w_act=None, bias=b[0], act=hip.ACT_SOFTPLUS100),
        dict(w_in=None, w_act=W[1], bias=b[1], act=hip.ACT_MUL_AUX),
        dict(w_in=Wx2, w_act=W[2], bias=b[2], act=hip.ACT_MUL2),
        dict(w_in=None, w_act=W[3], bias=b[3], act=hip.ACT_HEAD),
        dict(w_in=None, w_act=W[4], bias=b[4], act=hip.ACT_SOFTPLUS_BWD),
        dict(w_in=None, w_act=W[5], bias=b[5], act=hip.ACT_RELU_MASK),
        dict(w_in=None, w_act=W[6], bias=b[6], act=hip.ACT_RELU),
        dict(w_in=None, w_act=Wf, bias=bf, act=hip.ACT_NONE),
    ]
    d = lambda t: None if t is None else t.to(cuda)
    packed = fused.pack_layers([{k: (d(v) if torch.is_tensor(v) else v) for k, v in L.items()} for L in layers],
                               2, 0, 3, hip.OUT_NONE, cuda)
    save = [torch.empty(Q, 256, device=cuda) for _ in range(7)]
    save2 = [torch.empty(Q, 256, device=cuda) if i in (0, 1, 2) else None for i in range(7)] + [None]
    out = packed(d(x), Q, save=save, save2=save2,
                 mask=[None, d(m1), d(m2), None, d(s4), d(h5), None, None],
                 aux2=[None, None, d(q2), None, d(t4), None, None, None])
    D = lambda t: t.double()
    xd = D(x)
    z0 = xd @ D(W[0]).t() + D(b[0])
    a0 = torch.nn.functional.softplus(z0, beta=100)
    z1 = a0 @ D(W[1]).t() + D(b[1])
    a1 = z1 * D(m1)
    z2 = a1 @ D(W[2]).t() + xd @ D(Wx2).t() + D(b[2])
    a2 = z2 * D(m2)
    z3 = a2 @ D(W[3]).t() + D(b[3])
    z4 = a2 @ D(W[4]).t() + D(b[4])            # HEAD leaves the activations of layer 2 in place
    a4 = D(s4) * z4 + 100.0 * (1.0 - D(s4)) * D(t4)
    z5 = a4 @ D(W[5]).t() + D(b[5])
    a5 = torch.where(D(h5) > 0, z5, torch.zeros_like(z5))
    a6 = torch.relu(a5 @ D(W[6]).t() + D(b[6]))
    ref_out = a6 @ D(Wf).t() + D(bf)
    for name, got, ref in [('a0', save[0], a0), ('sigmoid(100 z0)', save2[0], torch.sigmoid(100 * z0)), ('a1', save[1], a1),
                           ('raw z1', save2[1], z1), ('a2', save[2], a2), ('z2 * q2', save2[2], z2 * D(q2)), ('head z3', save[3], z3),
                           ('a4', save[4], a4), ('a5', save[5], a5), ('a6', save[6], a6), ('out', out, ref_out)]:
        assert_close(got.cpu(), ref, 2e-5, 'chain program: ' + name)

    # a chain without final layer that starts from a tensor (act_init) plus a per-row init table
    act0, init = rn(Q, 256), rn(Q, 256)
    zeros = torch.zeros(256)
    ls = [dict(init_a=torch.zeros(256, 64), init_b=None, w_act=W[1], bias=zeros, act=hip.ACT_RELU_MASK),
          dict(w_act=W[2], bias=zeros, act=hip.ACT_NONE)]
    pk = fused.pack_layers([{k: (d(v) if torch.is_tensor(v) else v) for k, v in L.items()} for L in ls], 2, 0, 0,
                           hip.OUT_NONE, cuda, has_final=False)
    pk.init_wa = pk.init_wb = pk.init_bias = None
    pk.desc.init_stride = 256
    dumps = [torch.empty(Q, 256, device=cuda) for _ in range(2)]
    pk(None, Q, a_div=1, a_mod=Q, init_a_direct=d(init), act_init=d(act0), mask=[d(h5), None], save=dumps)
    r0 = D(act0) @ D(W[1]).t() + D(init)
    r0 = torch.where(D(h5) > 0, r0, torch.zeros_like(r0))
    assert_close(dumps[0].cpu(), r0, 2e-5, 'act_init + init table')
    assert_close(dumps[1].cpu(), r0 @ D(W[2]).t(), 2e-5, 'second chain layer')
    # act_init for a row PREFIX only (act_init_rows): the rows behind it start from zero activations -- bit-identical to
    # the launch that reads a zero-padded tensor
    for q1 in (1, 63, 64, Q - 1):
        pad = torch.cat([d(act0)[:q1], torch.zeros(Q - q1, 256, device=cuda)], 0)
        ref_d = [torch.empty(Q, 256, device=cuda) for _ in range(2)]
        pk(None, Q, a_div=1, a_mod=Q, init_a_direct=d(init), act_init=pad, mask=[d(h5), None], save=ref_d)
        got_d = [torch.empty(Q, 256, device=cuda) for _ in range(2)]
        pk(None, Q, a_div=1, a_mod=Q, init_a_direct=d(init), act_init=d(act0)[:q1].contiguous(), act_init_rows=q1,
           mask=[d(h5), None], save=got_d)
        assert torch.equal(got_d[0], ref_d[0]) and torch.equal(got_d[1], ref_d[1]), 'act_init_rows=%d' % q1


def test_gemm_tn_grouped_row_prefix_products(cuda):
    """PsnGemmTnItem.k_rows: a product of the group may sum over a row PREFIX of the pass (operands with fewer rows than the
    first item) -- on the 256 x 256, the 256 x 64 and the 128 x 128 tile paths, incl. prefixes that leave whole K slices empty."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(21)
    K = 9000
    mk = lambda r, c: torch.randn(r, c, generator=g).to(cuda)
    for k1 in (1, 500, 4097, K - 1):
        items = [dict(A=mk(K, 256), B=mk(K, 256), colsum=True),              # full pass, 256 x 256 tiles
                 dict(A=mk(k1, 256), B=mk(k1, 256), colsum=True),            # prefix, 256 x 256 tiles
                 dict(A=mk(k1, 200), B=mk(k1, 39), colsum=True),             # prefix, 256 x 64 tiles
                 dict(A=mk(k1, 96), B=mk(k1, 128), colsum=True),             # prefix, 128 x 128 tiles
                 dict(A=mk(K, 96), B=mk(K, 70))]
        res = hip.gemm_tn_grouped(items)
        for i, (it, (C, cs)) in enumerate(zip(items, res)):
            ref = it['A'].double().t() @ it['B'].double()
            assert_close(C.double().cpu(), ref.cpu(), 2e-5, 'k_rows=%d item %d' % (k1, i))
            if cs is not None:
                assert_close(cs.double().cpu(), it['A'].double().sum(0).cpu(), 2e-5, 'k_rows=%d item %d column sums' % (k1, i))


@pytest.mark.parametrize('with_outer,with_noise', [(False, False), (True, True), (False, True)])
def test_sample_points_bit_exact(cuda, with_outer, with_noise):
    """psn_sample_points against the torch formulation of stage1/model/rendering.py:110-176 (the reference's op order):
    bit-identical depths / points for hit rays (inner interval, optional outer samples) and miss rays, with and without
    stratified jitter, rows selected through index lists."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(11)
    N, steps, steps_out, near, delta = 777, 64, 32, 28.0, 0.35
    S = steps + steps_out if with_outer else steps
    cam = torch.randn(N, 3, generator=g).to(cuda)
    rays = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(cuda)
    far = (33.0 + torch.rand(N, generator=g)).to(cuda)
    dist = (29.0 + 3.0 * torch.rand(N, generator=g)).to(cuda)
    dist[:5] = near + 0.1  # dnp clamps to near
    hit = (torch.rand(N, generator=g) < 0.6).to(cuda)
    hit_idx, miss_idx = hit.nonzero(as_tuple=True)[0], (~hit).nonzero(as_tuple=True)[0]
    nz_m = torch.rand(miss_idx.numel(), S, generator=g).to(cuda) if with_noise else None
    nz_h = torch.rand(hit_idx.numel(), S, generator=g).to(cuda) if with_noise else None
    U = lambda n: (torch.linspace(0.0, 1.0, steps=n, device=cuda), 1.0 - torch.linspace(0.0, 1.0, steps=n, device=cuda))
    out = torch.zeros(N, S, 3, device=cuda)
    hip.sample_points(cam, rays, far, out, False, near, U(S), idx=miss_idx, noise=nz_m)
    if with_outer:
        hip.sample_points(cam, rays, far, out, True, near, U(steps_out), idx=hit_idx, dist=dist, delta=delta, u1=U(steps), noise=nz_h)
    else:
        hip.sample_points(cam, rays, far, out, True, near, U(steps), idx=hit_idx, dist=dist, delta=delta, noise=nz_h)

    def jitter(d, nz):
        mid = 0.5 * (d[:, 1:] + d[:, :-1])
        hi = torch.cat([mid, d[:, -1:]], dim=-1)
        lo = torch.cat([d[:, :1], mid], dim=-1)
        return lo + (hi - lo) * nz

    u = torch.linspace(0.0, 1.0, steps=S, device=cuda).view(1, -1)
    d2 = near * (1.0 - u) + far[miss_idx].view(-1, 1) * u
    if with_noise:
        d2 = jitter(d2, nz_m)
    ref = torch.zeros(N, S, 3, device=cuda)
    ref[miss_idx] = cam[miss_idx].unsqueeze(-2) + rays[miss_idx].unsqueeze(-2) * d2.unsqueeze(-1)
    dh, fh = dist[hit_idx], far[hit_idx]
    dnp, dfp = dh - delta, dh + delta
    dnp = torch.where(dnp < near, torch.full_like(dnp, near), dnp)
    dfp = torch.where(dfp > fh, fh, dfp)
    u = torch.linspace(0.0, 1.0, steps=steps, device=cuda).view(1, -1)
    d1 = dnp.view(-1, 1) * (1.0 - u) + dfp.view(-1, 1) * u
    if with_outer:
        uo = torch.linspace(0.0, 1.0, steps=steps_out, device=cuda).view(1, -1)
        d_out = near * (1.0 - uo) + dnp.view(-1, 1) * uo
        d1, _ = torch.sort(torch.cat([d_out, d1], dim=-1), dim=-1)
    if with_noise:
        d1 = jitter(d1, nz_h)
    ref[hit_idx] = cam[hit_idx].unsqueeze(-2) + rays[hit_idx].unsqueeze(-2) * d1.unsqueeze(-1)
    assert torch.equal(out, ref), 'max |diff| = %g' % float((out - ref).abs().max())


def test_weight_norm_all(cuda):
    """ops.WeightNormAll (csrc/weight_norm.hip) = v * (g / |v|_row) * scale and its autograd, all layers in one launch
    (nn.utils.weight_norm of stage1/model/network.py:37-66)."""
    from psnerf_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 39), (256, 256), (217, 256), (256, 256), (257, 256), (3, 256), (1, 5), (70, 289)]
    scales = (1.0, 1.0, 1.0, float(1 / np.sqrt(2)), 1.0, 1.0, 1.0, 0.5)
    vs = [torch.randn(s, generator=g).to(cuda).requires_grad_(True) for s in shapes]
    gs = [(torch.rand(s[0], 1, generator=g) + 0.5).to(cuda).requires_grad_(True) for s in shapes]
    cot = [torch.randn(s, generator=g).to(cuda) for s in shapes]
    gv = []
    for a, b in zip(gs, vs):
        gv += [a, b]
    Ws = ops.WeightNormAll.apply(scales, *gv)
    sum((w * c).sum() for w, c in zip(Ws[:-1], cot[:-1])).backward()  # the last output gets no gradient
    got = [(v.grad.clone(), a.grad.clone()) for v, a in zip(vs, gs)]
    for t in vs + gs:
        t.grad = None
    ref = [v * (a / v.norm(2, dim=1, keepdim=True)) * s if s != 1.0 else v * (a / v.norm(2, dim=1, keepdim=True))
           for v, a, s in zip(vs, gs, scales)]
    sum((w * c).sum() for w, c in zip(ref[:-1], cot[:-1])).backward()
    for i, (w, r) in enumerate(zip(Ws, ref)):
        assert_close(w.detach().cpu(), r.detach().cpu(), 2e-6, 'w[%d]' % i)
    for i, ((dv, dg), v, a) in enumerate(zip(got, vs, gs)):
        if i == len(vs) - 1:
            assert float(dv.abs().max()) == 0.0 and float(dg.abs().max()) == 0.0
        else:
            assert_close(dv.cpu(), v.grad.cpu(), 2e-5, 'dv[%d]' % i)
            assert_close(dg.cpu(), a.grad.cpu(), 2e-5, 'dg[%d]' % i)


def test_weight_norm_more_layers_than_one_launch(cuda):
    """hip.weight_norm_fwd / _bwd split lists longer than PSN_WN_MAX_ITEMS (16) into several launches."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(5)
    vs = [torch.randn(8 + i, 5 + 2 * i, generator=g).to(cuda) for i in range(19)]
    gs = [(torch.rand(v.shape[0], generator=g) + 0.5).to(cuda) for v in vs]
    sc = [1.0 + 0.1 * (i % 3) for i in range(19)]
    ws = hip.weight_norm_fwd(vs, gs, sc)
    dws = [torch.randn(v.shape, generator=g).to(cuda) for v in vs]
    dvs, dgs = hip.weight_norm_bwd(vs, gs, sc, dws)
    for v, a, s_, w, dw, dv, dg in zip(vs, gs, sc, ws, dws, dvs, dgs):
        v64, a64, dw64 = v.double(), a.double(), dw.double()
        nrm = v64.norm(dim=1, keepdim=True)
        assert_close(w.cpu(), (v64 * (a64[:, None] / nrm) * s_).float().cpu(), 2e-6, 'w')
        dot = (dw64 * v64).sum(1, keepdim=True) * s_
        assert_close(dg.cpu(), (dot / nrm).float().reshape(-1).cpu(), 2e-5, 'dg')
        assert_close(dv.cpu(), (dw64 * (a64[:, None] / nrm) * s_ - dot * a64[:, None] * v64 / nrm ** 3).float().cpu(), 2e-5, 'dv')


def test_secant_step_matches_torch_formulation(cuda):
    """hip.secant_step = one iteration of stage1/model/rendering.py:525-555 (bracket update, next estimate, next query
    point), bit-identical to the elementwise torch formulation."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(9)
    n, tau = 1000, 0.5
    d_low = (torch.rand(n, generator=g) * 2 + 1).to(cuda)
    d_high = d_low + (torch.rand(n, generator=g) * 0.1 + 1e-3).to(cuda)
    f_low = -(torch.rand(n, generator=g) * 0.5 + 1e-3).to(cuda)
    f_high = (torch.rand(n, generator=g) * 0.5 + 1e-3).to(cuda)
    origin = torch.randn(n, 3, generator=g).to(cuda)
    direction = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(cuda)
    ref = dict(d_low=d_low.clone(), d_high=d_high.clone(), f_low=f_low.clone(), f_high=f_high.clone())
    d_pred = torch.empty(n, device=cuda)
    p_mid = torch.empty(n, 3, device=cuda)
    hip.secant_step(None, tau, d_pred, d_low, d_high, f_low, f_high, origin, direction, p_mid)
    r_pred = -ref['f_low'] * (ref['d_high'] - ref['d_low']) / (ref['f_high'] - ref['f_low']) + ref['d_low']
    assert torch.equal(d_pred, r_pred)
    assert torch.equal(p_mid, origin + r_pred.unsqueeze(-1) * direction)
    for it in range(3):
        occ = torch.rand(n, generator=g).to(cuda)
        hip.secant_step(occ, tau, d_pred, d_low, d_high, f_low, f_high, origin, direction, p_mid if it < 2 else None)
        f_mid = occ - tau
        lo = f_mid < 0
        ref['d_low'] = torch.where(lo, r_pred, ref['d_low'])
        ref['f_low'] = torch.where(lo, f_mid, ref['f_low'])
        ref['d_high'] = torch.where(lo, ref['d_high'], r_pred)
        ref['f_high'] = torch.where(lo, ref['f_high'], f_mid)
        r_pred = -ref['f_low'] * (ref['d_high'] - ref['d_low']) / (ref['f_high'] - ref['f_low']) + ref['d_low']
        assert torch.equal(d_pred, r_pred), it
        for k, t in (('d_low', d_low), ('d_high', d_high), ('f_low', f_low), ('f_high', f_high)):
            assert torch.equal(t, ref[k]), (it, k)


def test_row_adam_kernel_matches_sparse_adam(cuda):
    """psn_row_adam (one launch for both light tables) = torch.optim.SparseAdam on the touched rows, several steps with
    changing row sets and duplicate rows; untouched rows and their moments do not move."""
    from psnerf_amd.optim import RowSparseAdam
    g = torch.Generator().manual_seed(0)
    n = 50
    w3, w1 = torch.randn(n, 3, generator=g), torch.randn(n, 1, generator=g)
    ref3, ref1 = torch.nn.Embedding(n, 3, sparse=True), torch.nn.Embedding(n, 1, sparse=True)
    a3, a1 = torch.nn.Embedding(n, 3).to(cuda), torch.nn.Embedding(n, 1).to(cuda)
    ref3.weight.data.copy_(w3); ref1.weight.data.copy_(w1); a3.weight.data.copy_(w3); a1.weight.data.copy_(w1)
    o_ref = torch.optim.SparseAdam([{'params': list(ref3.parameters())}, {'params': list(ref1.parameters()), 'lr': 1e-2}], lr=5e-3)
    o_a = RowSparseAdam([{'params': list(a3.parameters())}, {'params': list(a1.parameters()), 'lr': 1e-2}], lr=5e-3)
    for it in range(5):
        rows = torch.randint(0, n, (9,), generator=g)
        c3, c1 = torch.randn(9, 3, generator=g), torch.randn(9, 1, generator=g)
        o_ref.zero_grad(); o_a.zero_grad()
        ((ref3(rows) * c3).sum() + (ref1(rows) * c1).sum()).backward()
        ((a3(rows.to(cuda)) * c3.to(cuda)).sum() + (a1(rows.to(cuda)) * c1.to(cuda)).sum()).backward()
        o_ref.step()
        o_a.step(rows=rows.to(cuda))
        assert_close(a3.weight.detach().cpu(), ref3.weight.detach(), 1e-6, 'dir table it%d' % it, atol=1e-7)
        assert_close(a1.weight.detach().cpu(), ref1.weight.detach(), 1e-6, 'intensity table it%d' % it, atol=1e-7)
    st = o_a.state[a3.weight]
    assert int(st['step']) == 5 and set(st.keys()) == {'step', 'exp_avg', 'exp_avg_sq'}


@pytest.mark.parametrize('Q', [1, 257, 40000])
def test_app_input_table(cuda, Q):
    """psn_app_input == the torch formulation of the appearance network's input (network.py:128-138, 141-150): columns
    [p | gamma(v / |v|) | normal | 0], bands bit-identical to psn_pe_encode of the normalised direction."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(Q)
    p = torch.randn(Q, 3, generator=g).to(cuda)
    v = (torch.randn(Q, 3, generator=g) * 3.0).to(cuda)
    n = torch.randn(Q, 3, generator=g).to(cuda)
    x = hip.app_input(p, v, n, 4)
    vn = v / torch.norm(v, dim=-1, keepdim=True)
    pe = hip.pe_encode(vn.contiguous(), 4, 27, 1.0)
    assert x.shape == (Q, 64)
    assert torch.equal(x[:, :3], p) and torch.equal(x[:, 30:33], n) and torch.equal(x[:, 33:], torch.zeros(Q, 31, device=cuda))
    assert_close(x[:, 3:30].cpu(), pe.cpu(), 1e-6, 'view encoding', atol=1e-6)  # |v| by sqrtf(x^2 + y^2 + z^2) vs torch.norm: last bits
    ref = torch.cat([vn, *[f(vn * 2 ** k) for k in range(4) for f in (torch.sin, torch.cos)]], dim=-1)
    assert_close(x[:, 3:30].double().cpu(), ref.double().cpu(), 1e-5, 'view encoding vs torch', atol=1e-6)


def test_small_fused_row_kernels_vs_torch(cuda):
    """csrc/small.hip against the torch formulations they replace: F.normalize + its autograd backward on [n, 3] rows (incl. a
    zero row and a row below eps), the light-table lookups of stage2/trainer.py:376-379 with duplicate indices (dense table
    gradients), and the camera rays of rend_util.py:90-147 for selected pixels."""
    from psnerf_amd import hip, ops
    from psnerf_amd.stage2.renderer import camera_rays
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, 3, generator=g)
    x[3] = 0.0
    x[4] = 1e-14
    x[5] *= 1e3
    gy = torch.randn(1000, 3, generator=g)
    xr = x.clone().requires_grad_()
    yr = torch.nn.functional.normalize(xr, p=2, dim=-1)
    yr.backward(gy)
    xd = x.to(cuda).requires_grad_()
    yd = ops.normalize_rows(xd)
    yd.backward(gy.to(cuda))
    assert_close(yd.detach().cpu(), yr.detach(), 1e-6, 'normalize fwd', atol=1e-7)
    assert_close(xd.grad.cpu(), xr.grad, 1e-5, 'normalize bwd', atol=1e-6 * float(xr.grad.abs().max()))
    # light rows: duplicates, both tables
    NL, L = 40, 17
    dt, it = torch.randn(NL, 3, generator=g), torch.rand(NL, 1, generator=g) + 1.0
    idx = torch.randint(0, NL, (L,), generator=g)
    idx[5] = idx[2]
    idx[9] = idx[2]
    gd, gi = torch.randn(L, 3, generator=g), torch.randn(L, 1, generator=g)
    e1, e2 = torch.nn.Embedding(NL, 3), torch.nn.Embedding(NL, 1)
    e1.weight.data.copy_(dt)
    e2.weight.data.copy_(it)
    ((torch.nn.functional.normalize(e1(idx), p=2, dim=-1) * gd).sum() + (e2(idx) * gi).sum()).backward()
    dtd, itd = dt.to(cuda).requires_grad_(), it.to(cuda).requires_grad_()
    d, i = ops.LightRows.apply(dtd, itd, idx.to(cuda))
    ((d * gd.to(cuda)).sum() + (i * gi.to(cuda)).sum()).backward()
    assert_close(d.detach().cpu(), torch.nn.functional.normalize(dt[idx], dim=-1), 1e-6, 'light dir', atol=1e-7)
    assert torch.equal(i.detach().cpu(), it[idx])
    assert_close(dtd.grad.cpu(), e1.weight.grad, 1e-5, 'd dir table', atol=1e-6)
    assert_close(itd.grad.cpu(), e2.weight.grad, 1e-6, 'd intensity table', atol=1e-7)
    untouched = torch.ones(NL, dtype=torch.bool)
    untouched[idx] = False
    assert float(dtd.grad.cpu()[untouched].abs().max()) == 0.0
    # only one table needs a gradient
    dtd2 = dt.to(cuda).requires_grad_()
    d2, i2 = ops.LightRows.apply(dtd2, it.to(cuda), idx.to(cuda))
    (d2 * gd.to(cuda)).sum().backward()
    assert torch.equal(dtd2.grad, dtd.grad)
    # camera rays of a pixel subset, negated
    from psnerf_amd.synthetic import look_at_pose
    uv = torch.stack([torch.randint(0, 612, (500,), generator=g).float(), torch.randint(0, 512, (500,), generator=g).float()], -1)[None]
    K = torch.eye(4)[None].clone()
    K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2] = 3759.0, 3741.5, 306.0, 256.0
    pose = look_at_pose(31.5, az_deg=40.0, el_deg=-15.0)[None]
    sel = torch.sort(torch.randperm(500, generator=g)[:123]).values
    ref = -camera_rays(uv, pose, K)[0][0][sel]
    got = hip.camera_rays(uv.to(cuda), pose.to(cuda), K.to(cuda), sel.to(cuda), scale=-1.0)
    assert_close(got.cpu(), ref, 1e-6, 'camera rays', atol=1e-7)
    assert_close(hip.camera_rays(uv.to(cuda), pose.to(cuda), K.to(cuda)).cpu(), camera_rays(uv, pose, K)[0][0], 1e-6, 'camera rays (all)', atol=1e-7)


def test_flat_adam_follows_torch_adam(cuda):
    """optim.FlatAdam (one psn_adam_flat launch over flat parameter / moment / gradient buffers) against torch.optim.Adam on
    the same parameter set and gradients: several steps, a parameter that is frozen at first and joins later (its own step
    count), the MultiStepLR schedule, state_dict round trip into torch.optim.Adam and back."""
    from psnerf_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(128, 63), (128,), (3, 128), (3,), (64, 5), (7,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    pa = [t.clone().to(cuda).requires_grad_() for t in init]
    pb = [t.clone().to(cuda).requires_grad_() for t in init]
    oa = FlatAdam(pa, lr=5e-4)
    ob = torch.optim.Adam(pb, lr=5e-4, foreach=True)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, [3], gamma=0.5)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, [3], gamma=0.5)
    frozen = 4
    pa[frozen].requires_grad_(False)
    pb[frozen].requires_grad_(False)
    for it in range(6):
        if it == 2:
            pa[frozen].requires_grad_(True)
            pb[frozen].requires_grad_(True)
        grads = [torch.randn(s, generator=g).to(cuda) * (10.0 ** (it - 3)) for s in shapes]
        assert oa.attach_grads()
        ob.zero_grad()
        for p, q, gr in zip(pa, pb, grads):
            if p.requires_grad:
                assert p.grad is None
                p.grad = gr.clone()  # a tensor of its own, as autograd hands it over; step() gathers them (one multi-tensor copy)
                q.grad = gr.clone()
        assert pa[frozen].grad is None or it >= 2
        oa.step()
        assert all(p.grad is None or (p.grad._base is not None and p.grad._base.data_ptr() == pa[0].grad._base.data_ptr()) for p in pa)
        ob.step()
        sa.step()
        sb.step()
        for k, (p, q) in enumerate(zip(pa, pb)):
            assert_close(p.detach().cpu(), q.detach().cpu(), 1e-6, 'param %d after step %d' % (k, it), atol=1e-7)
    assert int(oa.state[pa[frozen]]['step']) == 4 and int(oa.state[pa[0]]['step']) == 6
    # parameters and moments are views of one allocation each; the state dict is interchangeable with torch.optim.Adam's
    base = pa[0].data_ptr()
    off = 0
    for p in pa:  # one allocation, every parameter on a 256-byte boundary
        assert p.data_ptr() == base + 4 * off and p.data_ptr() % 256 == 0
        off += (p.numel() + 63) // 64 * 64
    sd = oa.state_dict()
    assert all(t._base is None for st in sd['state'].values() for t in st.values() if torch.is_tensor(t))
    oc = torch.optim.Adam([t.detach().clone().requires_grad_() for t in pa], lr=1.0)
    oc.load_state_dict(sd)
    od = FlatAdam([t.detach().clone().requires_grad_() for t in pb], lr=1.0)
    od.load_state_dict(ob.state_dict())
    for (p1, p2) in zip(oc.param_groups[0]['params'], od.param_groups[0]['params']):
        g1 = torch.randn(p1.shape, generator=g).to(cuda)
        p1.grad = g1.clone()
        p2.grad = g1.clone()
    oc.step()
    od.step()
    for p1, p2 in zip(oc.param_groups[0]['params'], od.param_groups[0]['params']):
        assert_close(p1.detach().cpu(), p2.detach().cpu(), 1e-6, 'after state-dict exchange', atol=1e-7)


def test_flat_adam_state_dict_leaves_the_live_state_alone(cuda):
    """Optimizer.state_dict() hands out the LIVE per-parameter dictionaries; FlatAdam.state_dict() must
    clone into fresh ones.  step, state_dict(), step, state_dict(): the second checkpoint holds the moments of step 2 (equal
    to torch.optim.Adam's), the live moments still are views of the flat buffers, and a gradient that is a view of some
    unrelated 1-D tensor is copied like any loose gradient."""
    from psnerf_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(33, 7), (5,), (64, 64)]
    init = [torch.randn(s, generator=g) for s in shapes]
    pa = [t.clone().to(cuda).requires_grad_() for t in init]
    pb = [t.clone().to(cuda).requires_grad_() for t in init]
    oa, ob = FlatAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2, foreach=True)
    sds = []
    for it in range(3):
        oa.attach_grads()
        ob.zero_grad()
        foreign = torch.randn(sum(torch.Size(s).numel() for s in shapes) + 100, generator=g).to(cuda)  # NOT a registered flat buffer
        off = 17
        for p, q in zip(pa, pb):
            p.grad = foreign[off:off + p.numel()].view_as(p)  # a contiguous view of a 1-D tensor with arbitrary neighbours
            q.grad = p.grad.clone()
            off += p.numel()
        oa.step()
        ob.step()
        sds.append(oa.state_dict())
        for p in pa:
            st = oa.state[p]
            assert st['exp_avg']._base is not None and st['exp_avg']._base.data_ptr() == oa._flat['m'].data_ptr()
            assert st['exp_avg_sq']._base is not None and st['exp_avg_sq']._base.data_ptr() == oa._flat['v'].data_ptr()
    ref = ob.state_dict()
    for k, st in sds[-1]['state'].items():
        assert float(st['step']) == 3.0
        assert_close(st['exp_avg'].cpu(), ref['state'][k]['exp_avg'].cpu(), 1e-6, 'exp_avg %d' % k, atol=1e-7)
        assert_close(st['exp_avg_sq'].cpu(), ref['state'][k]['exp_avg_sq'].cpu(), 1e-6, 'exp_avg_sq %d' % k, atol=1e-9)
        assert not torch.equal(st['exp_avg'], sds[0]['state'][k]['exp_avg'])  # checkpoints are not frozen at the first one
    for p, q in zip(pa, pb):
        assert_close(p.detach().cpu(), q.detach().cpu(), 1e-6, 'params', atol=1e-7)
    # alignment gaps of all flat buffers still hold zeros
    f = oa._flat
    live = torch.zeros_like(f['p'], dtype=torch.bool)
    for p in pa:
        live[f['off'][p]:f['off'][p] + p.numel()] = True
    for name in ('p', 'm', 'v'):
        assert float(f[name][~live].abs().max()) == 0.0, name


@pytest.mark.parametrize('h,w', [(512, 612), (23, 37), (64, 64)])
def test_stage1_targets_equal_torch_formulation(cuda, h, w):
    """psn_stage1_targets against the torch formulation of training.py:166-191 (five nearest grid_samples, the angle mask on
    the unrotated normal, the rotation): pure gathers and exact products -> identical bits, every pixel of the image
    (incl. the x = w/2 tie of even widths) plus out-of-range positions."""
    from psnerf_amd import hip
    from psnerf_amd.stage1.training import Trainer
    g = torch.Generator().manual_seed(h * 1000 + w)
    img = torch.rand(1, 3, h, w, generator=g).to(cuda)
    mask = (torch.rand(1, 1, h, w, generator=g) > 0.4).float().to(cuda)
    valid = (torch.rand(1, 1, h, w, generator=g) > 0.1).float().to(cuda)
    nmask = (torch.rand(1, 1, h, w, generator=g) > 0.3).float().to(cuda)
    normal = torch.nn.functional.normalize(torch.randn(1, 3, h, w, generator=g), dim=1).to(cuda)
    world = torch.eye(4).unsqueeze(0)
    world[0, :3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    world = world.to(cuda)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).float()
    extra = torch.tensor([[-1.0, 0.0], [w * 1.0, 3.0], [2.0, h * 1.0], [w + 5.0, -3.0], [0.4, 0.6], [w - 1.4, h - 1.5]])
    pix = torch.cat([pix, extra]).unsqueeze(0).to(cuda).contiguous()
    for angle, want_normal in ((None, True), (70.0, True), (70.0, False)):
        t = Trainer.__new__(Trainer)
        t.normal_loss, t.angle = True, angle
        ref = t._targets_torch(img, mask, valid, normal, nmask, world, pix, want_normal)
        cos_t = float(np.cos(np.deg2rad(angle))) if (angle is not None and want_normal) else None
        got = hip.stage1_targets(pix[0], img[0], mask[0, 0], valid[0, 0], normal[0] if want_normal else None, nmask[0, 0],
                                 world[0] if want_normal else None, cos_t, want_normal)
        rgb, mask_gt, mask_valid, norm_mask_gt, normal_gt = got
        assert torch.equal(rgb, ref[0][0]) and torch.equal(mask_gt, ref[1][0]) and torch.equal(mask_valid, ref[2][0])
        assert mask_valid.dtype == torch.bool and norm_mask_gt.dtype == torch.bool
        assert torch.equal(norm_mask_gt, ref[3][0])
        if want_normal:
            assert torch.equal(normal_gt, ref[4][0])
            assert 0 < int(norm_mask_gt.sum()) < norm_mask_gt.numel()
        else:
            assert normal_gt is None and ref[4] is None
    # absent mask images are all ones
    rgb, mask_gt, mask_valid, norm_mask_gt, normal_gt = hip.stage1_targets(pix[0], img[0])
    assert bool(mask_gt.min() == 1) and bool(mask_valid.all()) and norm_mask_gt is None and normal_gt is None


def test_stage1_rays_and_surface_points_vs_torch_formulation(cuda):
    """psn_stage1_rays / psn_surface_points against the torch formulations they replace (rendering.py pixel_rays,
    sphere_intersection, _march_finish + _surface) and against the oracle on the CPU."""
    from psnerf_amd import hip
    from psnerf_amd.stage1 import rendering as R
    from oracle import stage1 as o1
    g = torch.Generator().manual_seed(5)
    n, h, w = 5000, 512, 612
    pix = torch.stack([torch.randint(0, w, (n,), generator=g), torch.randint(0, h, (n,), generator=g)], -1).float().unsqueeze(0)
    K = torch.eye(4).unsqueeze(0)
    K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2] = 1200.0, 1190.0, 306.0, 256.0
    ang = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    W = torch.eye(4).unsqueeze(0)
    W[0, :3, :3] = ang
    W[0, :3, 3] = -ang[:, 2] * 3.0 + torch.tensor([0.05, -0.02, 0.1])  # camera 3 units out, looking roughly at the origin
    for radius in (2.0, 1.0, 0.1):  # 0.1: most rays miss the sphere
        cam_t = R.camera_origin(n, W.to(cuda))
        rays_t = R.pixel_rays(pix.to(cuda), K.to(cuda), W.to(cuda))
        rays_t = rays_t / rays_t.norm(2, 2).unsqueeze(-1)
        far_t, hit_t = R.sphere_intersection(cam_t[:, 0], rays_t, r=radius)
        cam, rays, far = hip.stage1_rays(pix[0].to(cuda), K[0].to(cuda), W[0].to(cuda), radius)
        cam3, rays3, far3 = hip.stage1_rays(pix[0].to(cuda), K[0, :3, :3].contiguous().to(cuda), W[0].to(cuda), radius)
        assert torch.equal(cam, cam_t[0]) and torch.equal(cam, cam3) and torch.equal(rays, rays3) and torch.equal(far, far3)
        assert_close(rays.cpu(), rays_t[0].cpu(), 1e-6, 'rays')
        assert int(((far > 0) != (far_t[0, :, 1] > 0)).sum()) <= 2  # grazing rays may flip with the last bit of b
        if radius == 0.1:
            assert int((far == 0).sum()) > n // 2
        # the oracle's chain on the CPU (float64: the kernel's rounding against the exact value)
        rays_o = o1.pixel_rays(pix.double(), K.double(), W.double())
        rays_o = rays_o / rays_o.norm(2, 2).unsqueeze(-1)
        far_o, hit_o = o1.sphere_intersection(W[:, :3, 3].double(), rays_o, r=radius)
        assert_close(rays.cpu(), rays_o[0].float(), 1e-6, 'rays vs oracle')
        both = (far.cpu() > 0) & hit_o[0]
        assert int(((far.cpu() > 0) != hit_o[0]).sum()) <= 2
        # far = sqrt(under) - b with under = b^2 - (|c|^2 - r^2) a difference of O(10) numbers: fp32 rounding of under (~4e-6)
        # reaches far as 4e-6 / (2 sqrt(under)) -- the bound follows the conditioning (grazing rays), for the kernel and for
        # the torch formulation alike
        c64 = W[0, :3, 3].double()
        b64 = (rays_o[0] * c64).sum(-1)
        under64 = (b64 ** 2 - (c64.norm() ** 2 - radius ** 2))[both]
        bound = 1e-5 + 4e-6 / under64.sqrt()
        for name, f in (('kernel', far.cpu()), ('torch formulation', far_t[0, :, 1].cpu())):
            err = (f[both].double() - far_o[0, :, 1][both]).abs()
            assert bool((err <= bound).all()), '%s: far off by %.2e x its conditioning bound' % (name, float((err / bound).max()))
    # surface points: every flag combination, crossing depths incl. 0, inf, nan
    d_pred = torch.rand(4096, generator=g) * 3
    d_pred[::17] = 0.0
    d_pred[5::19] = float('inf')
    d_pred[7::23] = float('nan')
    flags = torch.randint(0, 4, (4096,), generator=g, dtype=torch.int32)
    cam4, rays4 = torch.randn(4096, 3, generator=g), torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1)
    d_pred, flags, cam4, rays4 = (t.to(cuda) for t in (d_pred, flags, cam4, rays4))
    out = torch.where((flags & 1).bool(), d_pred, torch.full_like(d_pred, float('inf')))
    d_ref = torch.where((flags & 2).bool(), out, torch.zeros_like(out))
    zero_occ, ok = d_ref == 0, R.finite_mask(d_ref)
    dists_ref = torch.where(zero_occ, torch.zeros_like(d_ref), torch.where(ok, d_ref, torch.ones_like(d_ref)))
    dists, obj, pts, d_i = hip.surface_points(d_pred, flags, cam4, rays4, want_d=True)
    assert torch.equal(torch.nan_to_num(d_i, nan=-7.0), torch.nan_to_num(d_ref, nan=-7.0))
    assert torch.equal(dists, dists_ref) and torch.equal(obj, ok & ~zero_occ) and obj.dtype == torch.bool
    assert torch.equal(pts, cam4 + rays4 * dists_ref.unsqueeze(-1))


def _stage1_loss_inputs(n, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    rgb, rgb_gt = r(n, 3), r(n, 3)
    gfield = torch.randn(2 * n, 3, generator=g)
    gfield[n:] = gfield[:n] + 0.05 * torch.randn(n, 3, generator=g)
    gfield[3] = 0.0            # |g| = 0: the norm's gradient is 0 there
    gfield[n + 5] = gfield[5]  # identical normals: |a - b| = 0, its gradient 0
    hit = r(n) > 0.3
    normal_gt = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    norm_mask = r(n) > 0.5
    acc = r(n) * 1.2 - 0.1     # some values outside [0, 1]: clamp, zero gradient there
    acc[7], acc[8] = 0.0, 1.0  # log terms clamped at -100
    mask_gt = (r(n) > 0.5).float()
    mask_valid = r(n) > 0.2
    return [t.to(dev) for t in (rgb, rgb_gt, gfield, hit, normal_gt, norm_mask, acc, mask_gt, mask_valid)]


@pytest.mark.parametrize('n,terms_on', [(4096, 'all'), (777, 'all'), (4096, 'rgb_grad'), (300, 'no_mask'), (64, 'empty_masks')])
def test_stage1_fused_losses_and_surface_normals_vs_torch_formulation(cuda, n, terms_on):
    """ops.SurfaceNormals + ops.Stage1Losses (csrc/loss1.hip) against the torch formulation they replace (rendering.py:200-212
    as written in Renderer._unisurf_sync_free, losses.py:24-70 as written in Loss.forward): every loss term and the gradients
    with respect to colours, field gradients and accumulated opacity."""
    from psnerf_amd import ops
    from psnerf_amd.stage1 import Loss
    rgb, rgb_gt, gfield, hit, normal_gt, norm_mask, acc, mask_gt, mask_valid = _stage1_loss_inputs(n, cuda, seed=n)
    if terms_on == 'empty_masks':
        hit, norm_mask, mask_valid = torch.zeros_like(hit), torch.zeros_like(norm_mask), torch.zeros_like(mask_valid)
    with_norm, with_mask = terms_on in ('all', 'empty_masks'), terms_on in ('all', 'empty_masks')
    if terms_on == 'no_mask':
        with_norm = True
    res = {}
    for fused in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (rgb, gfield, acc)]
        c, gf, a = leaves
        if fused:
            norm_pred, diff = ops.SurfaceNormals.apply(gf, hit)
        else:
            nrm = gf / (gf.norm(2, dim=1).unsqueeze(-1) + 10 ** (-5))
            norm_pred = torch.where(hit.unsqueeze(-1), nrm[:n], torch.zeros_like(nrm[:n]))
            diff = torch.norm(nrm[:n] - nrm[n:], dim=-1)
        out = {'rgb': c.reshape(1, n, 3), 'mask_pred': hit, 'diff_norm': None, 'diff_norm_full': diff,
               'normal_pred': norm_pred.reshape(1, n, 3), 'acc_map': a.reshape(1, n)}
        loss = Loss(1.0, 0.005, 0.05, 1.0, device=cuda)
        loss.fused = fused
        terms = loss(out, rgb_gt.reshape(1, n, 3), normal_gt.reshape(1, n, 3) if with_norm else None, norm_mask.reshape(1, n),
                     a.reshape(1, n) if with_mask else None, mask_gt.reshape(1, n) if with_mask else None, mask_valid.reshape(1, n))
        terms['loss'].backward()
        res[fused] = ({k: float(v.detach()) for k, v in terms.items()}, [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in leaves],
                      norm_pred.detach(), diff.detach())
    assert set(res[True][0]) == set(res[False][0])
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 2e-6 * max(abs(v), 1e-3), '%s: %r vs %r' % (k, res[True][0][k], v)
    assert_close(res[True][2].cpu(), res[False][2].cpu(), 1e-6, 'normal_pred')
    assert float((res[True][3] - res[False][3]).abs().max()) <= 5e-7, 'diff_norm'
    for name, a_, b_ in zip(('d rgb', 'd field gradient', 'd acc'), res[True][1], res[False][1]):
        assert torch.isfinite(a_).all()
        assert_close(a_.cpu(), b_.cpu(), 2e-5, name)
    if terms_on == 'empty_masks':
        assert res[True][0]['grad_loss'] == 0.0 and res[True][0]['normal_loss'] == 0.0 and res[True][0]['mask_loss'] == 0.0


def test_stage1_fused_losses_global_counts_hook(cuda):
    """reduce_counts (the data-parallel hook) sees the three mask counts and its result is what the terms and the gradients
    divide by: doubling the counts halves the masked terms and their gradients."""
    from psnerf_amd import ops
    n = 512
    rgb, rgb_gt, gfield, hit, normal_gt, norm_mask, acc, mask_gt, mask_valid = _stage1_loss_inputs(n, cuda, seed=1)
    seen = {}

    def double(t):
        seen['counts'] = t.clone()
        t.mul_(2.0)
        return t
    outs = []
    diff0, normal0 = torch.rand(n, device=cuda), torch.randn(n, 3, device=cuda)
    for hook in (None, double):
        a = acc.clone().requires_grad_(True)
        diff, normal = diff0.clone().requires_grad_(True), normal0.clone().requires_grad_(True)
        loss, terms = ops.Stage1Losses.apply(rgb, rgb_gt, diff, hit, normal, normal_gt, norm_mask, a, mask_gt, mask_valid, 2 * n,
                                             (1.0, 0.5, 0.25, 2.0), hook)
        loss.backward()
        outs.append((terms.clone(), diff.grad.clone(), normal.grad.clone(), a.grad.clone()))
    assert seen['counts'].tolist() == [float(hit.sum()), float(norm_mask.sum()), float(mask_valid.sum())]
    t0, t1 = outs[0][0], outs[1][0]
    assert float(t0[0]) == float(t1[0])  # colour term: divided by n_rays, not by a count
    for i in (1, 2, 3):
        assert abs(float(t1[i]) * 2 - float(t0[i])) <= 1e-6 * abs(float(t0[i]))
    for a_, b_ in zip(outs[0][1:], outs[1][1:]):
        assert_close((b_ * 2).cpu(), a_.cpu(), 1e-6, 'gradient under doubled counts')


def test_softplus100_accuracy_against_float64(cuda):
    """The device softplus(beta = 100) + sigmoid (csrc/common.h softplus100_pair: packed fp32 pairs, hardware exp2 / log2, no
    select for the threshold) against float64, next to torch's OWN fp32 softplus on the same inputs: the same error level
    (both are dominated by the rounding of 100 z), exactly z above the threshold, and -- through the occupancy engine, which
    uses the value-only form -- agreement of the two forms."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(0)
    z = torch.cat([torch.rand(1 << 18, generator=g) * 1.3 - 0.8, torch.rand(1 << 18, generator=g) * 0.1 - 0.05,
                   torch.randn(1 << 17, generator=g), torch.tensor([0.0, 0.2, 0.20000002, 0.19999999, -0.2, 1e-8, -1e-8, 3.0, -0.86])])
    n = z.numel() // 64 * 64
    z = z[:n].reshape(-1, 64).contiguous().to(cuda)
    eye = torch.eye(64, device=cuda)
    s_out = torch.empty_like(z)
    sp = hip.gemm(z, eye, trans_b=True, bias=torch.zeros(64, device=cuda), epi=hip.EPI_BIAS_SOFTPLUS, aux_out=s_out)
    z64 = z.double()
    ref = torch.where(z64 * 100 > 20, z64, torch.log1p(torch.exp(torch.clamp(z64 * 100, max=50.0))) / 100)
    ours = ((sp.double() - ref).abs() / ref)
    theirs = ((torch.nn.functional.softplus(z, beta=100).double() - ref).abs() / ref)
    ok = ref > 1e-36  # below: exp underflows in fp32, in torch as here
    assert float(ours[ok].max()) <= 1.15 * float(theirs[ok].max()) + 1e-7, (float(ours[ok].max()), float(theirs[ok].max()))
    assert float(ours[ok].mean()) <= 1.15 * float(theirs[ok].mean()) + 1e-8
    above = z * 100 > 20
    assert torch.equal(sp[above], z[above])  # torch returns x itself above the threshold; so does max(x, 0) + l / 100
    sig_ref = torch.sigmoid(z64 * 100)
    k = sig_ref > 1e-36
    their_sig = ((torch.sigmoid(z * 100).double() - sig_ref).abs() / sig_ref)[k].max()
    assert float(((s_out.double() - sig_ref).abs() / sig_ref)[k].max()) <= 1.15 * float(their_sig) + 1e-7


@pytest.mark.parametrize('n', [1, 63, 1024, 4097, 262144])
def test_mask_count_and_inverse_index_equal_torch(cuda, n):
    """psn_mask_count == (a & b).sum() as a float; psn_inverse_index == fill(-1) + index_put(arange) on an ascending index list
    (the front of a stage-2 step: four + three torch launches -> one each)."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(n)
    a = (torch.rand(n, generator=g) < 0.9).to(cuda)
    b = (torch.rand(n, generator=g) < 0.7).to(cuda)
    assert float(hip.mask_count(a, b)) == float((a & b).sum())
    assert float(hip.mask_count(a)) == float(a.sum())
    assert hip.mask_count(a, b).dtype == torch.float32 and hip.mask_count(a, b).shape == (1,)
    if n > 9:  # unaligned views (the byte path) and masks whose true bytes are not 1 (a uint8 buffer viewed as bool)
        assert float(hip.mask_count(a[1:], b[1:])) == float((a[1:] & b[1:]).sum())
        raw = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8).to(cuda)
        raw[::3] = 0
        assert float(hip.mask_count(raw.view(torch.bool), a)) == float(((raw != 0) & a).sum())
    idx = a.nonzero(as_tuple=True)[0]
    inv = hip.inverse_index(idx, n)
    ref = torch.full((n,), -1, dtype=torch.int32, device=cuda)
    ref[idx] = torch.arange(idx.numel(), dtype=torch.int32, device=cuda)
    assert torch.equal(inv, ref)
    assert torch.equal(hip.inverse_index(idx[:0], n), torch.full((n,), -1, dtype=torch.int32, device=cuda))


@pytest.mark.parametrize('V,Ns,n_items', [(8, 3655, 2), (3, 17, 1), (16, 1000, 2), (1, 64, 2)])
def test_pair_sums_group_vs_torch(cuda, V, Ns, n_items):
    """psn_pair_sums_group (the separable input-block gradients of ops.VisibilityPair.backward, every input layer in two launches)
    against the torch formulation: sum over the lights, (sum over the points)^T PE(l), and the bias sum."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(V * 1000 + Ns)
    xs = [torch.randn(V * Ns, 256, generator=g).to(cuda) for _ in range(n_items)]
    pe = torch.randn(V, 64, generator=g).to(cuda)
    pe[:, 63] = 0
    res = hip.pair_sums_group(xs, V, Ns, pe, 64, [i == 0 for i in range(n_items)])
    for i, (x, (sx, dWl, b)) in enumerate(zip(xs, res)):
        x3 = x.double().view(V, Ns, 256)
        assert_close(sx.cpu(), x3.sum(0).float().cpu(), 1e-6, 'sx', atol=1e-5)
        dzl = x3.sum(1)
        assert_close(dWl.cpu(), (dzl.t() @ pe.double()).float().cpu(), 1e-5, 'dWl')
        if i == 0:
            assert_close(b.cpu(), dzl.sum(0).float().cpu(), 1e-5, 'bias')
        else:
            assert b is None


def test_copy_group_copies_any_dtype_and_alignment(cuda):
    """psn_copy_bytes_group: contiguous tensors of mixed element types, odd byte counts and unaligned views in one launch."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(0)
    base = torch.randn(100000, generator=g).to(cuda)
    srcs = [torch.randn(1, 4096, 3, generator=g).to(cuda), (torch.rand(1, 4097, generator=g) < 0.5).to(cuda), torch.randint(0, 1 << 40, (3001,), generator=g).to(cuda),
            base[1:50002], torch.randn(96, 1234, 3, generator=g).to(cuda), torch.randint(0, 255, (13,), generator=g, dtype=torch.uint8).to(cuda),
            torch.randn(1, generator=g).to(cuda)]
    pad = torch.zeros(100001, device=cuda)
    dsts = [torch.zeros_like(s) for s in srcs]
    dsts[3] = pad[3:50004]  # (4-byte aligned only, both sides)
    hip.copy_group(list(zip(dsts, srcs)) * 4)  # 28 items: two launches
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    assert float(pad[:3].abs().sum()) == 0.0 and float(pad[50004:].abs().sum()) == 0.0  # nothing written outside the views
