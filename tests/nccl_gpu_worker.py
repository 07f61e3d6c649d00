#!/usr/bin/env python3
"""RCCL on the one GPU of the test box: a process group with backend 'nccl' (= RCCL on ROCm) and world_size 1.

    python tests/nccl_gpu_worker.py OUT.json

In a fresh process (started by tests/conftest.py before pytest initialises HIP): torch.cuda.set_device, init_process_group
('nccl'), then one stage-2 ``TrainStep`` and two stage-1 ``Trainer`` steps through psnerf_amd.dist.DataParallel(force=True)
-- the data-parallel code path with its device-resident count all-reduces and the flat-bucket gradient all-reduce issued on
the current HIP stream, every collective executed by RCCL (identities in a world of one) -- and the same steps without data
parallelism: losses, gradients and updated parameters must be IDENTICAL (bit for bit: a one-rank SUM changes nothing).
Also times the bucket all-reduce.  This is the only RCCL evidence obtainable without a multi-GPU node.
Writes {"ok": bool, ...} to OUT.json.
"""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    res = {'ok': False}
    try:
        import torch
        import torch.distributed as dist
        from psnerf_amd import dist as pdist
        rank, local, world = pdist.init_from_env(backend='nccl', set_device=True, force=True)
        dev = torch.device('cuda', local)
        res['backend'] = dist.get_backend()
        assert res['backend'] == 'nccl' and dist.get_world_size() == 1
        checks = {}

        # ---- stage 2 -------------------------------------------------------------------------------------------------
        import psnerf_amd.stage2 as s2
        from psnerf_amd.synthetic import stage2_inputs, stage1_batch, stage1_cfg
        from tests.helpers import stage1_state_dict, stage2_state_dict
        from oracle import stage2 as o2  # only for the seeded state dict of the parity tests
        conf = s2.bear_conf()
        sd = stage2_state_dict(o2.bear_conf(), seed=5)
        N, L, V, NL = 1000, 5, 3, 24
        g = torch.Generator().manual_seed(3)
        light_init = torch.nn.functional.normalize(torch.randn(NL, 3, generator=g), dim=-1)
        l_slt = torch.randperm(NL, generator=g)[:L].to(dev)
        inp, gt = stage2_inputs(N, L, V, seed=17, device=dev)
        nz = (torch.randn(int(inp['surface_mask'].sum()), 3, generator=g) * 0.01).to(dev)
        runs = []
        for force in (True, False):
            dp = pdist.DataParallel(dev, force=force)
            assert dp.enabled == force
            net = s2.PSNetwork(conf)
            net.load_state_dict(sd)
            net.to(dev)
            st = s2.TrainStep(net, conf, NL, light_init.to(dev), dev, dp=dp)
            st.cur_iter = 5001
            terms, _ = st.step(inp, gt, l_slt, train_order=False, noise={'xyz': nz})
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            grads['__light_dir'] = st.light_para.weight.grad.detach().clone()
            grads['__light_int'] = st.light_inten_para.weight.grad.detach().clone()
            runs.append((dp, {k: float(v.detach()) for k, v in terms.items() if v is not None}, grads,
                         {k: v.detach().clone() for k, v in net.state_dict().items()}))
        (dp1, t1, g1, p1), (_, t0, g0, p0) = runs
        assert dp1.n_allreduce == 1 and dp1.allreduce_bytes > 2_000_000, (dp1.n_allreduce, dp1.allreduce_bytes)
        checks['stage2 loss terms identical'] = t1 == t0
        checks['stage2 gradients identical'] = sorted(g1) == sorted(g0) and all(torch.equal(g1[k], g0[k]) for k in g0)
        checks['stage2 parameters after the step identical'] = all(torch.equal(p1[k], p0[k]) for k in p0)
        res['stage2_bucket_bytes'] = dp1.allreduce_bytes
        res['allreduce_ms_bucket'] = round(dp1.time_allreduce(dp1.allreduce_bytes // 4), 4)

        # ---- stage 2 replayed from HIP graphs AROUND the RCCL collectives (stage2/graph.py: two graphs per step, the count and
        # the bucket all-reduce issued eagerly between them on the caller's stream) against the eager data-parallel step ------
        from psnerf_amd.stage2.graph import GraphedTrainStep
        g2 = torch.Generator().manual_seed(8)
        nzs = [(torch.randn(int(inp['surface_mask'].sum()), 3, generator=g2) * 0.01).to(dev) for _ in range(7)]
        gres = []
        for graphed in (False, True):
            dp = pdist.DataParallel(dev, force=True)
            net = s2.PSNetwork(conf)
            net.load_state_dict(sd)
            net.to(dev)
            st = s2.TrainStep(net, conf, NL, light_init.to(dev), dev, dp=dp)
            st.cur_iter = 5001
            run = GraphedTrainStep(st, warmup=2) if graphed else st
            losses = []
            for k in range(7):
                terms, _ = run.step(inp, gt, l_slt, train_order=False, noise={'xyz': nzs[k]})
                losses.append(float(terms['total'].detach()))
            torch.cuda.synchronize()
            if graphed:
                assert run.n_captures == 1 and run.n_eager == 2 and run.n_replays == 5, (run.n_captures, run.n_eager, run.n_replays)
                assert len(run._captured[next(iter(run._captured))].graphs) == 2  # [fwd, bwd, gather] + [optimisers]
            assert dp.n_allreduce == 7, dp.n_allreduce
            gres.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()},
                         st.light_para.weight.detach().clone(), st.light_inten_para.weight.detach().clone()))
        (le, pe_, ae, be), (lg, pg, ag, bg) = gres
        checks['stage2 graphed data-parallel steps: losses identical'] = le == lg
        checks['stage2 graphed data-parallel steps: parameters and light tables identical'] = (
            all(torch.equal(pe_[k], pg[k]) for k in pe_) and torch.equal(ae, ag) and torch.equal(be, bg))

        # ---- stage 1 (sync-free forward: device-resident counts go through RCCL as tensors) ---------------------------
        from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
        n_rays = 160
        cfg = stage1_cfg('bunny', **{'training.n_training_points': n_rays})
        sd1 = stage1_state_dict(cfg, seed=21)
        batch = {k: v.to(dev) for k, v in stage1_batch(cfg, h=48, w=64, seed=4).items()}
        runs = []
        for force in (True, False):
            net = NeuralNetwork(cfg)
            net.load_state_dict(sd1)
            dp = pdist.DataParallel(dev, force=force)
            tr = Trainer(Renderer(net, cfg, device=dev), torch.optim.Adam(net.parameters(), lr=1e-4), cfg, device=dev, dp=dp)
            losses = []
            for it in (1500, 1501):
                gen = torch.Generator().manual_seed(it)
                pix = torch.stack([torch.randint(0, 64, (n_rays,), generator=gen).float(),
                                   torch.randint(0, 48, (n_rays,), generator=gen).float()], -1)[None]
                nzs = {'full': torch.rand(n_rays, 64, generator=gen).to(dev), 'nbr_full': torch.rand(n_rays, 3, generator=gen).to(dev)}
                losses.append({k: float(v.detach()) for k, v in tr.train_step(batch, it=it, pix=pix, noise=nzs).items()})
            runs.append((dp, losses, {k: p.grad.detach().clone() for k, p in net.named_parameters()},
                         {k: v.detach().clone() for k, v in net.state_dict().items()}))
        (dp1, l1, g1, p1), (_, l0, g0, p0) = runs
        assert dp1.n_allreduce == 2
        checks['stage1 loss terms identical'] = l1 == l0
        checks['stage1 gradients identical'] = all(torch.equal(g1[k], g0[k]) for k in g0)
        checks['stage1 parameters after two steps identical'] = all(torch.equal(p1[k], p0[k]) for k in p0)
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        res.update(ok=all(checks.values()), checks=checks)
    except Exception as e:  # noqa: BLE001
        import traceback
        res['error'] = '%s\n%s' % (e, traceback.format_exc())
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    sys.exit(0 if res['ok'] else 1)


if __name__ == '__main__':
    main()
