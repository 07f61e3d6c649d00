"""The stage-2 train step replayed from HIP graphs (psnerf_amd/stage2/graph.py) against the eager step: a graph changes who
issues the launches, never what they compute -- losses, parameters, light tables and optimiser states must be IDENTICAL."""
import numpy as np
import pytest
import torch

from tests.helpers import stage2_state_dict
from psnerf_amd.synthetic import stage2_inputs

pytestmark = pytest.mark.gpu


def _batches(n_it, N, L, V, NL, dev, same_masks=True):
    """Batches of one geometry (the surface count must not change: it is a tensor SHAPE of the step) with different contents."""
    base_inp, base_gt = stage2_inputs(N, L, V, seed=100)
    out = []
    for it in range(n_it):
        inp, gt = stage2_inputs(N, L, V, seed=200 + it)
        for k in ('surface_mask', 'object_mask'):
            inp[k] = base_inp[k].clone()
        ns = int(inp['surface_mask'].sum())
        inp['surface_idx'] = inp['surface_mask'][0].nonzero(as_tuple=True)[0]
        l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L]
        nz = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
        out.append(({k: v.to(dev) for k, v in inp.items()}, {k: v.to(dev) for k, v in gt.items()}, l_slt.to(dev), {'xyz': nz.to(dev)}))
    return out


def _make(cuda, start_iter, overlap=True):
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    sd = stage2_state_dict(conf, seed=9)
    NL = 40
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    net = s2.PSNetwork(conf)
    net.load_state_dict(sd)
    net.to(cuda)
    net.overlap_small_nets = overlap
    step = s2.TrainStep(net, conf, NL, light_init.to(cuda), cuda, milestones=[start_iter + 4])
    if start_iter > 0:
        # the state train_fix left at iteration 0 (as in test_train_steps_match_oracle)
        step.cur_iter = start_iter
        step._ori = (1.0, 0.05, 0.01, 1)
        step.loss.sg_rgb_weight, step.loss.albedo_smooth_weight, step.loss.rough_smooth_weight, step.loss.vis_weight = 0, 0, 0, 10
        step.model.albedo_net.eval().requires_grad_(False)
        step.model.rough_net.eval().requires_grad_(False)
        step.light_para.requires_grad_(False)
        step.light_inten_para.requires_grad_(False)
        if start_iter > 5000:
            step.cur_iter = 5000
            step.train_fix()
            step.cur_iter = start_iter
    return step, NL


def _state(step):
    sd = {k: v.detach().cpu().clone() for k, v in step.model.state_dict().items()}
    sd['__light_dir'] = step.light_para.weight.detach().cpu().clone()
    sd['__light_int'] = step.light_inten_para.weight.detach().cpu().clone()
    for name, opt in (('sg', step.sg_optimizer), ('light', step.light_optimizer)):
        for i, st in opt.state_dict()['state'].items():
            for k, v in st.items():
                sd['__opt_%s_%s_%s' % (name, i, k)] = v.detach().cpu().clone() if torch.is_tensor(v) else torch.tensor(float(v))
    return sd


@pytest.mark.parametrize('overlap', [True, False])
@pytest.mark.parametrize('start_iter', [4995, 5001])
def test_graphed_step_is_bit_identical_to_eager(cuda, start_iter, overlap):
    """Ten optimisation steps with changing batch contents, an lr milestone inside and (start 4995) the train_fix switch at
    iteration 5000 in the middle -- eager TrainStep.step against GraphedTrainStep (2 eager warm-up steps per signature, then a
    capture, then replays): losses of every step, final parameters, light tables, Adam / SparseAdam states bit for bit."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    N, L, V, n_it = 3000, 12, 4, 10
    res = {}
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, start_iter, overlap)
        batches = _batches(n_it, N, L, V, NL, cuda)
        run = GraphedTrainStep(step, warmup=2) if mode == 'graph' else step
        losses = []
        for inp, gt, l_slt, nz in batches:
            terms, out = run.step(inp, gt, l_slt, train_order=True, noise=nz)
            losses.append(torch.stack([terms[k].detach().reshape(()).clone() for k in ('total', 'sg_rgb_loss', 'vis_loss', 'normal_loss')]))
        torch.cuda.synchronize()
        if mode == 'graph':
            n_sig = 2 if start_iter < 5000 else 1
            assert run.n_captures == n_sig and run.n_eager == 2 * n_sig and run.n_replays == n_it - 2 * n_sig, \
                (run.n_captures, run.n_eager, run.n_replays)
        assert step.cur_iter == start_iter + n_it
        res[mode] = (torch.stack(losses).cpu(), _state(step), out['sg_rgb_values'].detach().cpu().clone())
    (l_e, s_e, o_e), (l_g, s_g, o_g) = res['eager'], res['graph']
    assert torch.isfinite(l_e).all()
    assert torch.equal(l_e, l_g), (l_e - l_g).abs().max()
    assert s_e.keys() == s_g.keys()
    for k in s_e:
        assert torch.equal(s_e[k], s_g[k]), k
    assert torch.equal(o_e, o_g)


def test_alternating_signatures_and_an_empty_batch_keep_every_graph_on_its_own_plan(cuda):
    """Two batch geometries (different pixel counts -> two graphs with different surface counts) replayed alternately, with an
    UNPADDED batch without a single surface pixel in between (its loss has no graph: every .grad is None, both optimisers have
    nothing to do -- and must neither crash beside live graphs nor wipe the host-side plans the replays advance their step counts
    and step sizes from; optim.*.graph_advance(plan), one plan snapshot per capture).  Against the eager step: bit for bit."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    res = {}
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, 5001)
        big = _batches(6, 3000, 12, 4, NL, cuda)
        small = _batches(6, 1700, 12, 4, NL, cuda)
        e_inp, e_gt, e_l, _ = _batches(1, 1700, 12, 4, NL, cuda)[0]
        e_inp = dict(e_inp)
        e_inp['surface_mask'] = torch.zeros_like(e_inp['surface_mask'])
        e_inp['surface_idx'] = e_inp['surface_idx'][:0]
        order = [big[0], small[0], big[1], small[1], big[2], small[2], (e_inp, e_gt, e_l, None), big[3], small[3],
                 (e_inp, e_gt, e_l, None), small[4], big[4], big[5], small[5]]
        run = GraphedTrainStep(step, warmup=2) if mode == 'graph' else step
        losses = []
        for inp, gt, l_slt, nz in order:
            terms, out = run.step(inp, gt, l_slt, train_order=True, noise=nz)
            losses.append(terms['total'].detach().reshape(()).clone())
        torch.cuda.synchronize()
        if mode == 'graph':
            # the empty batch is a third signature: it runs eagerly both times (warm-up count 2), the two real ones are captured
            assert run.n_captures == 2 and run.n_eager == 6 and run.n_replays == len(order) - 6, (run.n_captures, run.n_eager, run.n_replays)
        res[mode] = (torch.stack(losses).cpu(), _state(step))
    (l_e, s_e), (l_g, s_g) = res['eager'], res['graph']
    assert torch.isfinite(l_e).all() and torch.equal(l_e, l_g), (l_e - l_g).abs().max()
    for k in s_e:
        assert torch.equal(s_e[k], s_g[k]), k
    # the step counts advanced with the steps that HAD gradients only: 12 of the 14
    steps = [float(v) for k, v in s_g.items() if k.startswith('__opt_sg_') and k.endswith('_step')]
    assert steps and all(v == 12.0 for v in steps), steps


def test_row_sparse_adam_without_gradient_leaves_no_plan_behind(cuda):
    """a step in which no table has a gradient issues no launch; with graph scalars on, the plan a capture would
    snapshot must then be EMPTY -- not the plan of an earlier fused step, whose replays would advance the tables' step counts
    (and with them the Adam bias corrections of later real steps) without an update."""
    from psnerf_amd.optim import RowSparseAdam, StepScalars
    table = torch.nn.Parameter(torch.randn(12, 3, device=cuda))
    opt = RowSparseAdam([table], lr=1e-2)
    opt.graph_scalars = StepScalars(4, cuda)
    rows = torch.tensor([1, 5, 5, 7], device=cuda)
    table.grad = torch.zeros_like(table)
    table.grad[rows] = 1.0
    opt.step(rows)
    assert len(opt._graph_plan) == 1 and int(opt.state[table]['step']) == 1
    table.grad = None
    opt.step(rows)
    assert opt._graph_plan == [] and int(opt.state[table]['step']) == 1
    opt.graph_advance(list(opt._graph_plan))   # what a replay of a graph captured around that step does: nothing
    assert int(opt.state[table]['step']) == 1


def test_graphed_step_draws_fresh_jitter_noise_every_replay(cuda):
    """Without injected noise the model draws the xyz jitter on the device (renderer.py:212): under replay the generator's
    offset must advance, i.e. two replays of the same batch see different jitter and therefore different smoothness terms."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    step, NL = _make(cuda, 5001)
    inp, gt, l_slt, _ = _batches(1, 2000, 6, 3, NL, cuda)[0]
    run = GraphedTrainStep(step, warmup=1)
    vals = []
    for _ in range(5):
        # (learning rates to zero: the parameters stay put, so that only the noise can change the jitter outputs)
        for g in step.sg_optimizer.param_groups + step.light_optimizer.param_groups:
            g['lr'] = 1e-30
        terms, out = run.step(inp, gt, l_slt, train_order=False)
        vals.append(out['albedo_jitter'].detach().clone())
    torch.cuda.synchronize()
    assert run.n_replays >= 3
    assert not torch.equal(vals[-1], vals[-2]) and not torch.equal(vals[-2], vals[-3])


@pytest.mark.parametrize('n', [1, 1000, 4097, 70000])
def test_surface_index_is_nonzero_padded_with_the_last_entry(cuda, n):
    """psn_surface_index: ascending positions of the set mask bytes, the tail of the fixed-size list repeating the last one; an
    empty mask gives zeros; psn_inverse_index maps a pixel to the FIRST of equal entries."""
    from psnerf_amd import hip
    g = torch.Generator().manual_seed(n)
    for frac in (0.9, 0.0, 1.0):
        m = (torch.rand(n, generator=g) < frac).to(cuda)
        idx, cnt = hip.surface_index(m, n)
        ref = m.nonzero(as_tuple=True)[0]
        k = ref.numel()
        assert float(cnt) == float(k)
        assert torch.equal(idx[:k], ref)
        assert bool((idx[k:] == (ref[-1] if k else 0)).all())
        want = torch.full((n,), -1, dtype=torch.int32, device=cuda)
        want[ref] = torch.arange(k, dtype=torch.int32, device=cuda)
        if k:
            assert torch.equal(hip.inverse_index(idx, n), want)
        # with the device-side count only the real entries map: an EMPTY mask gives no pixel a row (without it pixel 0 would get row 0)
        assert torch.equal(hip.inverse_index(idx, n, count=cnt), want)


@pytest.mark.parametrize('ns,live', [(1000, 517), (1024, 0), (1024, 1024), (4096, 3700), (640, 64)])
def test_padded_visibility_launch_skips_only_padding_workgroups(cuda, ns, live):
    """psn_mlp_infer_padded through ops.VisibilityPair.launch: with a device-side live count the shading rows (L groups of ns) of
    workgroups that hold padding only are zeros, every other output and every dump of the supervision rows is bit-identical to
    the plain launch."""
    from psnerf_amd import ops
    L, V = 5, 2
    g = torch.Generator().manual_seed(ns + live)
    pe_x = torch.randn(ns, 64, generator=g).to(cuda)
    pe_l = torch.randn(L + V, 64, generator=g).to(cuda)
    dims = [(256, 78)] + [(256, 256)] * 2 + [(256, 256 + 78)] + [(256, 256)] + [(1, 256)]
    params = []
    for o, i in dims:
        params += [(torch.randn(o, i, generator=g) / i ** 0.5).to(cuda), (torch.randn(o, generator=g) * 0.1).to(cuda)]
    c = torch.arange(39)
    cols = torch.cat([c, 64 + c]).to(cuda)
    ref, (save_ref, bits_ref) = ops.VisibilityPair.launch(pe_x, pe_l, L, cols, 2, params, True)
    cnt = torch.tensor([float(live)], device=cuda)
    out, (save, bits) = ops.VisibilityPair.launch(pe_x, pe_l, L, cols, 2, params, True, live_count=cnt)
    torch.cuda.synchronize()
    r = torch.arange((L + V) * ns, device=cuda)
    # a 64-row block of a shading light's group that lies behind the real rows is not evaluated (capacities that are no multiple
    # of 64 take the plain launch: a block would straddle two lights)
    dead = (r < L * ns) & (r % ns >= (live + 63) // 64 * 64) if ns % 64 == 0 else torch.zeros_like(r, dtype=torch.bool)
    assert int(dead.sum()) == (L * (ns - (live + 63) // 64 * 64) if ns % 64 == 0 else 0)
    assert bool((out[dead] == 0).all())
    assert torch.equal(out[~dead], ref[~dead])
    for a, b in zip(save + bits, save_ref + bits_ref):
        assert torch.equal(a, b)
    # the sign-bit words beside the dumps: bit 4 mt + r of word (row, g) = (feature 16 mt + 4 g + r of the dumped activation > 0)
    h = save[1].view(-1, 16, 4, 4)  # [row, mt, g, r]
    want = ((h > 0).long() << (4 * torch.arange(16, device=cuda).view(1, 16, 1, 1) + torch.arange(4, device=cuda).view(1, 1, 1, 4))).sum(dim=(1, 3))
    assert torch.equal(bits[1], want)


@pytest.mark.parametrize('ns,live,L,V', [(1000, None, 5, 2), (64, None, 3, 1), (4099, None, 13, 3), (29, None, 4, 2), (1024, 517, 5, 2),
                                         (4096, 3700, 11, 3), (640, 64, 5, 2), (1024, 0, 5, 2), (1024, 1024, 9, 8)])
def test_point_major_block_order_is_bit_identical_to_row_order(cuda, ns, live, L, V):
    """psn_mlp_block_order: the (light, point) row set of the visibility launch visited point-tile-major (light fastest, an eighth
    of the order per XCD) -- plain launches with group sizes that are no multiple of 64 (blocks straddle lights; groups shorter
    than a block fall back to row order) and padded launches with a device-side live count.  Every output row, every dump and
    every sign-bit word equals the row-order launch bit for bit (the row-order results stay alive, so the second launch cannot
    inherit their memory, and the block it is likely to get is poisoned with NaN first: a block that was left out shows)."""
    from psnerf_amd import ops, hip
    g = torch.Generator().manual_seed(ns * 7 + L)
    pe_x = torch.randn(ns, 64, generator=g).to(cuda)
    pe_l = torch.randn(L + V, 64, generator=g).to(cuda)
    dims = [(256, 78)] + [(256, 256)] * 2 + [(256, 256 + 78)] + [(256, 256)] + [(1, 256)]
    params = []
    for o, i in dims:
        params += [(torch.randn(o, i, generator=g) / i ** 0.5).to(cuda), (torch.randn(o, generator=g) * 0.1).to(cuda)]
    c = torch.arange(39)
    cols = torch.cat([c, 64 + c]).to(cuda)
    cnt = None if live is None else torch.tensor([float(live)], device=cuda)
    res = {}
    for order in ('row', 'point'):
        poison = torch.full(((L + V) * ns, 1), float('nan'), device=cuda)
        del poison
        with hip.block_order(order):
            out, (save, bits) = ops.VisibilityPair.launch(pe_x, pe_l, L, cols, 2, params, True, live_count=cnt)
        torch.cuda.synchronize()
        res[order] = [out] + save + bits
    assert not bool(torch.isnan(res['point'][0]).any())
    for a, b in zip(res['row'], res['point']):
        assert torch.equal(a, b)


@pytest.mark.parametrize('with_count', [False, True])
def test_padded_surface_list_gives_the_unpadded_step(cuda, with_count):
    """A surface-pixel list padded to the pixel count (dead rows = the last surface pixel repeated): the dense outputs of the
    forward are IDENTICAL to the unpadded step's, the losses too, and every gradient agrees to rounding (the dead rows add exact
    zeros; split-K chunks fall differently).  with_count: the batch also carries the device-side surface count, with which the
    visibility launch leaves the padding's shading rows out (zeros, dropped by the scatter)."""
    from psnerf_amd import hip
    res = {}
    for pad in (False, True):
        step, NL = _make(cuda, 5001)
        inp, gt, l_slt, nz = _batches(1, 3000, 12, 4, NL, cuda)[0]
        ns = int(inp['surface_mask'].sum())
        if pad:
            inp = dict(inp)
            inp['surface_idx'], cnt = hip.surface_index(inp['surface_mask'][0].contiguous(), 3000)
            if with_count:
                inp['surface_count'] = cnt
            nzp = torch.zeros(3000, 3, device=cuda)
            nzp[:ns] = nz['xyz']
            nz = {'xyz': nzp}
        terms, out, _, _ = step._fwd_bwd(inp, gt, l_slt, noise=nz)
        grads = {k: p.grad.detach().clone() for k, p in step.model.named_parameters() if p.grad is not None}
        grads['__light_dir'] = step.light_para.weight.grad.detach().clone()
        grads['__light_int'] = step.light_inten_para.weight.grad.detach().clone()
        res[pad] = ({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, {k: float(v.detach()) for k, v in terms.items() if v is not None}, grads)
    (o0, t0, g0), (o1, t1, g1) = res[False], res[True]
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k
    for k in t0:
        assert abs(t0[k] - t1[k]) <= 1e-6 * abs(t0[k]), (k, t0[k], t1[k])
    assert sorted(g0) == sorted(g1)
    from tests.helpers import assert_close
    for k in g0:
        assert_close(g1[k].cpu(), g0[k].cpu(), 1e-5, 'grad ' + k)


def test_one_graph_serves_batches_with_different_surface_counts(cuda):
    """pad_to_pixels: batches whose surface masks (and counts) differ replay ONE captured graph, built from the reference's
    dictionary (no 'surface_idx': the list is formed on the device); every step's losses agree with the eager, unpadded step
    from the same state."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    N, L, V, n_it = 2500, 10, 4, 8
    res = {}
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, 5001)
        run = GraphedTrainStep(step, warmup=1, pad_to_pixels=True) if mode == 'graph' else step
        losses, counts = [], []
        for it in range(n_it):
            inp, gt = stage2_inputs(N, L, V, seed=400 + it, surface_frac=0.6 + 0.04 * it)  # a different mask every step
            ns = int(inp['surface_mask'].sum())
            counts.append(ns)
            l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L].to(cuda)
            nz = torch.zeros(N, 3)
            nz[:ns] = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
            noise = {'xyz': (nz if mode == 'graph' else nz[:ns]).to(cuda)}
            terms, _ = run.step({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt, train_order=False, noise=noise)
            losses.append(float(terms['total'].detach()))
        res[mode] = losses
        if mode == 'graph':
            assert run.n_captures == 1 and run.n_eager == 1 and run.n_replays == n_it - 1, (run.n_captures, run.n_eager, run.n_replays)
    assert len(set(counts)) == n_it
    le, lg = np.array(res['eager']), np.array(res['graph'])
    assert np.isfinite(lg).all()
    rel = np.abs(lg - le) / np.abs(le)
    assert rel[0] <= 1e-6 and rel.max() <= 2e-3, rel  # (one step from a common state: equal; then two fp32 trajectories)


def test_pad_multiple_shares_a_graph_per_capacity(cuda):
    """pad_multiple = 512: batches that bring their surface list along are padded to the next multiple of 512 (the count is known
    on the host: the list's length); one graph per capacity, every step's loss equal to the eager, unpadded step's."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    N, L, V, n_it = 2560, 10, 4, 9
    res = {}
    caps = set()
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, 5001)
        run = GraphedTrainStep(step, warmup=1, pad_multiple=512) if mode == 'graph' else step
        losses = []
        for it in range(n_it):
            inp, gt = stage2_inputs(N, L, V, seed=500 + it, surface_frac=(0.55, 0.57, 0.75, 0.56, 0.76, 0.58, 0.74, 0.55, 0.77)[it])
            inp['surface_idx'] = inp['surface_mask'][0].nonzero(as_tuple=True)[0]
            ns = int(inp['surface_idx'].numel())
            cap = -(-ns // 512) * 512
            caps.add(cap)
            l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L].to(cuda)
            nz = torch.zeros(cap, 3)
            nz[:ns] = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
            noise = {'xyz': (nz if mode == 'graph' else nz[:ns]).to(cuda)}
            terms, _ = run.step({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt, train_order=False, noise=noise)
            losses.append(float(terms['total'].detach()))
        res[mode] = losses
        if mode == 'graph':
            assert len(caps) == 2, caps
            assert run.n_captures == 2 and run.n_eager == 2 and run.n_replays == n_it - 2, (run.n_captures, run.n_eager, run.n_replays)
    le, lg = np.array(res['eager']), np.array(res['graph'])
    assert np.isfinite(lg).all()
    rel = np.abs(lg - le) / np.abs(le)
    assert rel[0] <= 1e-6 and rel.max() <= 2e-3, rel


def test_padded_graph_with_an_empty_surface_mask(cuda):
    """A batch WITHOUT surface pixels through the padded graph (the list is n x pixel 0, the count on the device is 0): no pixel
    gets a row, so the dense outputs are the constant fills of the eager step on the same batch, and nothing is trained by it --
    the parameters after the step equal those of the eager step."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    N, L, V = 2048, 8, 4
    res = {}
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, 5001)
        run = GraphedTrainStep(step, warmup=1, pad_to_pixels=True) if mode == 'graph' else step
        for it in range(4):
            inp, gt = stage2_inputs(N, L, V, seed=600 + it, surface_frac=(0.8, 0.7, 0.0, 0.75)[it])
            if it == 2:
                inp['object_mask'] = torch.zeros_like(inp['object_mask'])
            l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L].to(cuda)
            ns = int(inp['surface_mask'].sum())
            nz = torch.zeros(N, 3)
            nz[:ns] = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
            noise = {'xyz': (nz if mode == 'graph' else nz[:ns]).to(cuda)}
            terms, out = run.step({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt, train_order=False, noise=noise)
            if it == 2:
                res[mode] = ({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, float(terms['total'].detach()),
                             {k: v.detach().clone() for k, v in step.model.state_dict().items()})
        if mode == 'graph':
            assert run.n_replays >= 2
    (o_e, t_e, _), (o_g, t_g, _) = res['eager'], res['graph']
    assert t_e == t_g == 0.0 or abs(t_e - t_g) <= 1e-7
    for k in o_e:
        if k in o_g and o_e[k].shape == o_g[k].shape:
            assert torch.equal(o_e[k], o_g[k]), k


def test_reset_drops_the_graphs_and_training_continues_identically(cuda):
    """GraphedTrainStep.reset() (what a caller does after load_stage2 / anything that replaces tensors the captured launches
    address): the next steps run eagerly and are captured again; the trajectory stays the eager one, bit for bit."""
    from psnerf_amd.stage2.graph import GraphedTrainStep
    N, L, V, n_it = 1500, 8, 4, 10
    res = {}
    for mode in ('eager', 'graph'):
        step, NL = _make(cuda, 5001)
        batches = _batches(n_it, N, L, V, NL, cuda)
        run = GraphedTrainStep(step, warmup=1) if mode == 'graph' else step
        losses = []
        for i, (inp, gt, l_slt, nz) in enumerate(batches):
            if mode == 'graph' and i == 5:
                assert run.n_captures == 1
                run.reset()
            terms, _ = run.step(inp, gt, l_slt, train_order=True, noise=nz)
            losses.append(terms['total'].detach().reshape(()).clone())
        torch.cuda.synchronize()
        if mode == 'graph':
            assert run.n_captures == 2 and run.n_eager == 2 and run.n_replays == n_it - 2, (run.n_captures, run.n_eager, run.n_replays)
        res[mode] = (torch.stack(losses).cpu(), _state(step))
    assert torch.equal(res['eager'][0], res['graph'][0])
    for k in res['eager'][1]:
        assert torch.equal(res['eager'][1][k], res['graph'][1][k]), k
