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
