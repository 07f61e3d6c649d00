"""Data-parallel path on CPU: two gloo ranks (world_size 2) vs one process.

The product kernels need a GPU, so the COMPUTE in these tests is the CPU oracle (test infrastructure);
what is under test is psnerf_amd.dist + the DP-aware losses: pixel sharding, global loss
denominators, the single flat-bucket all-reduce incl. the sparse light-table gradients.  The summed
rank gradients must equal the single-process gradients."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import ROOT


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from psnerf_amd import dist as pdist
    pdist.init_from_env(backend='gloo')
    torch.set_num_threads(2)
    from oracle import stage2 as o2
    from psnerf_amd.stage2.loss import MainLoss, NormalLoss
    from psnerf_amd.synthetic import stage2_inputs
    from tests.helpers import stage2_state_dict
    dp = pdist.DataParallel(torch.device('cpu'))
    assert dp.enabled and dp.world == world
    conf = o2.bear_conf()
    net = o2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=3))
    N, L, V, NL = 301, 3, 2, 10  # odd N: ragged shards
    inp, gt = stage2_inputs(N, L, V, seed=6)
    nz_full = torch.randn(int(inp['surface_mask'].sum()), 3, generator=torch.Generator().manual_seed(1)) * 0.01
    light_para = torch.nn.Embedding(NL, 3, sparse=True)
    light_int = torch.nn.Embedding(NL, 1, sparse=True)
    torch.manual_seed(0)
    light_para.weight.data.copy_(torch.nn.functional.normalize(torch.randn(NL, 3), dim=-1))
    torch.nn.init.constant_(light_int.weight, 2.0)
    l_slt = torch.tensor([7, 2, 5])
    mi, g = dp.shard_stage2(inp, gt)
    lo, hi = dp.slice_bounds(N)
    # the jitter noise rows of this rank's surface pixels
    surf_idx = torch.cumsum(inp['surface_mask'][0].long(), 0) - 1
    sel = inp['surface_mask'][0][lo:hi]
    nz = nz_full[surf_idx[lo:hi][sel]]
    mi['light_direction'] = torch.nn.functional.normalize(light_para(l_slt), p=2, dim=-1)
    mi['light_intensity'] = light_int(l_slt)
    out = net(mi, noise={'xyz': nz})
    ml, nl = MainLoss(1.0, 'L1', 0.05, 0.01, 1), NormalLoss(1, 0.05)
    ml.global_count = nl.global_count = dp.global_count
    loss = ml(out, g, mi)['loss'] + nl(out)['loss']
    loss.backward()
    dp.allreduce_grads(list(net.parameters()), [light_para.weight, light_int.weight])
    if rank == 0:
        torch.save({'grads': {k: p.grad for k, p in net.named_parameters()},
                    'light': light_para.weight.grad.to_dense(), 'lint': light_int.weight.grad.to_dense()}, tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    from oracle import stage2 as o2
    from psnerf_amd.synthetic import stage2_inputs
    from tests.helpers import stage2_state_dict, assert_close
    tmp = str(tmp_path / 'dp.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    # single-process reference
    conf = o2.bear_conf()
    net = o2.PSNetwork(conf)
    net.load_state_dict(stage2_state_dict(conf, seed=3))
    N, L, V, NL = 301, 3, 2, 10
    inp, gt = stage2_inputs(N, L, V, seed=6)
    nz_full = torch.randn(int(inp['surface_mask'].sum()), 3, generator=torch.Generator().manual_seed(1)) * 0.01
    light_para = torch.nn.Embedding(NL, 3, sparse=True)
    light_int = torch.nn.Embedding(NL, 1, sparse=True)
    torch.manual_seed(0)
    light_para.weight.data.copy_(torch.nn.functional.normalize(torch.randn(NL, 3), dim=-1))
    torch.nn.init.constant_(light_int.weight, 2.0)
    l_slt = torch.tensor([7, 2, 5])
    inp['light_direction'] = torch.nn.functional.normalize(light_para(l_slt), p=2, dim=-1)
    inp['light_intensity'] = light_int(l_slt)
    out = net(inp, noise={'xyz': nz_full})
    loss = o2.MainLoss(1.0, 'L1', 0.05, 0.01, 1)(out, gt, inp)['loss'] + o2.NormalLoss(1, 0.05)(out)['loss']
    loss.backward()
    for k, p in net.named_parameters():
        if p.grad is None:
            assert got['grads'][k] is None
            continue
        assert_close(got['grads'][k], p.grad, 2e-5, 'dp grad ' + k)
    assert_close(got['light'], light_para.weight.grad.to_dense(), 2e-5, 'dp light-dir grad')
    assert_close(got['lint'], light_int.weight.grad.to_dense(), 2e-5, 'dp light-intensity grad')


def test_slice_bounds_cover_everything_once():
    from psnerf_amd.dist import DataParallel
    dp = DataParallel()
    for world in (1, 2, 3, 8):
        dp.world = world
        for n in (0, 1, 7, 64, 1001):
            seen = []
            for r in range(world):
                dp.rank = r
                lo, hi = dp.slice_bounds(n)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


# ---- stage 1: ray sharding + global denominators of the stage-1 loss -----------------------------------------
def _stage1_outputs(theta, x):
    """A tiny differentiable stand-in for the renderer's output dictionary over rays x [1,N,3]: what is under
    test is the DP plumbing around it (shard_rays, the loss's global denominators, the flat-bucket all-reduce)."""
    y = x @ theta
    hit = x[0, :, 0] > 0
    return {'rgb': torch.sigmoid(y), 'diff_norm': (y[0][hit] ** 2).sum(-1), 'mask_pred': hit,
            'normal_pred': torch.nn.functional.normalize(y, dim=-1), 'acc_map': torch.sigmoid(y.sum(-1))}


def _stage1_case(n=203):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, n, 3, generator=g)
    x[0, 150:, 0] = -x[0, 150:, 0].abs()  # the second shard gets (almost) no hits: empty diff_norm on one rank
    rgb_gt = torch.rand(1, n, 3, generator=g)
    normal_gt = torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1)
    norm_mask = torch.rand(1, n, generator=g) > 0.4
    mask_gt = (torch.rand(1, n, generator=g) > 0.5).float()
    mask_valid = torch.rand(1, n, generator=g) > 0.2
    theta0 = torch.randn(3, 3, generator=g) * 0.5
    return x, rgb_gt, normal_gt, norm_mask, mask_gt, mask_valid, theta0


def _worker_stage1(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from psnerf_amd import dist as pdist
    from psnerf_amd.stage1.losses import Loss
    pdist.init_from_env(backend='gloo')
    dp = pdist.DataParallel(torch.device('cpu'))
    x, rgb_gt, normal_gt, norm_mask, mask_gt, mask_valid, theta0 = _stage1_case()
    theta = theta0.clone().requires_grad_()
    sh = dp.shard_rays  # every per-ray array is sliced the same way as the sampled pixel list
    out = _stage1_outputs(theta, sh(x))
    loss = Loss(1.0, 0.1, 0.5, 0.7)
    loss.global_sum = dp.global_sum_int
    loss.global_rays = lambda: dp.global_rays  # what the DP Trainer wires up: the pre-shard ray count, no collective
    terms = loss(out, sh(rgb_gt), sh(normal_gt), sh(norm_mask), out['acc_map'], sh(mask_gt), sh(mask_valid))
    terms['loss'].backward()
    dp.allreduce_grads([theta])
    grad1 = theta.grad.clone()
    # A second batch whose GLOBAL size differs while rank 0's LOCAL size stays the same (203 -> 204 rays: 102 / 101, then
    # 102 / 102): the rgb denominator used to be cached per local count with the all-reduce skipped on a hit -- rank 0
    # returned the stale 203 without a collective while rank 1 entered one (a hang).  Now no collective at all.
    x2, rgb2, ngt2, nm2, mg2, mv2, _ = _stage1_case(204)
    theta.grad = None
    out2 = _stage1_outputs(theta, sh(x2))
    t2 = loss(out2, sh(rgb2), sh(ngt2), sh(nm2), out2['acc_map'], sh(mg2), sh(mv2))
    t2['loss'].backward()
    dp.allreduce_grads([theta])
    if rank == 0:
        torch.save({'grad': grad1, 'grad2': theta.grad}, tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_stage1_two_rank_loss_gradients(tmp_path):
    from psnerf_amd.stage1.losses import Loss
    from tests.helpers import assert_close
    tmp = str(tmp_path / 'dp1.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_stage1, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    x, rgb_gt, normal_gt, norm_mask, mask_gt, mask_valid, theta0 = _stage1_case()
    theta = theta0.clone().requires_grad_()
    out = _stage1_outputs(theta, x)
    terms = Loss(1.0, 0.1, 0.5, 0.7)(out, rgb_gt, normal_gt, norm_mask, out['acc_map'], mask_gt, mask_valid)
    terms['loss'].backward()
    assert_close(got['grad'], theta.grad, 2e-5, 'stage-1 dp grad')
    x, rgb_gt, normal_gt, norm_mask, mask_gt, mask_valid, _ = _stage1_case(204)
    theta = theta0.clone().requires_grad_()
    out = _stage1_outputs(theta, x)
    Loss(1.0, 0.1, 0.5, 0.7)(out, rgb_gt, normal_gt, norm_mask, out['acc_map'], mask_gt, mask_valid)['loss'].backward()
    assert_close(got['grad2'], theta.grad, 2e-5, 'stage-1 dp grad, second batch of a different global size')


# ---- flat gradient bucket: one gather, views afterwards, a rank without a graph ------------------------------------------
def _worker_bucket(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from psnerf_amd import dist as pdist
    pdist.init_from_env(backend='gloo')
    dp = pdist.DataParallel(torch.device('cpu'))
    g = torch.Generator().manual_seed(0)
    ps = [torch.randn(5, 3, generator=g).requires_grad_(), torch.randn(7, generator=g).requires_grad_(),
          torch.randn(2, 2, generator=g)]  # the last one is frozen (requires_grad False): never in the bucket
    log = []
    for step in range(3):
        train = [p for p in ps if p.requires_grad]
        dp.prepare_grads(train)
        flat, views = dp._bucket(train)[:2]
        assert all(p.grad is None for p in train)  # autograd hands over its gradient tensors (no add_ per parameter)
        # rank 1 has "no surface pixel" on step 1: its loss has no graph and it skips backward (trainer behaviour)
        if not (rank == 1 and step == 1):
            loss = ((ps[0] * (rank + 1 + step)).sum() + (ps[1] ** 2).sum() * (rank + 1))
            loss.backward()
        dp.allreduce_grads(train)
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(train, views)), 'gradients are views of the reduced bucket'
        log.append([p.grad.clone() for p in train])
    assert dp.n_allreduce == 3 and ps[2].grad is None
    # a parameter that is frozen AFTER it received an all-reduced gradient must not keep it (an optimiser would go on
    # stepping on the stale view; zero_grad() sets it to None in the non-DP path): the trainers hand prepare_grads ALL
    # their parameters, it drops the gradients of the frozen ones
    ps[1].requires_grad_(False)
    dp.prepare_grads(ps)
    assert ps[1].grad is None and ps[0].grad is None and ps[2].grad is None
    (ps[0] * 2.0).sum().backward()
    dp.allreduce_grads([p for p in ps if p.requires_grad])
    assert torch.allclose(ps[0].grad, torch.full_like(ps[0], 2.0 * world))
    if rank == 0:
        torch.save({'log': log, 'ps': [p.detach() for p in ps]}, tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_views_and_rank_without_graph(tmp_path):
    tmp = str(tmp_path / 'bucket.pt')
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_bucket, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    ps = got['ps']
    for step, (g0, g1) in enumerate(got['log']):
        ranks = [0] if step == 1 else [0, 1]
        want0 = sum(float(r + 1 + step) for r in ranks) * torch.ones_like(ps[0])
        want1 = sum(2.0 * (r + 1) for r in ranks) * ps[1]
        assert torch.allclose(g0, want0) and torch.allclose(g1, want1, rtol=1e-6), step
