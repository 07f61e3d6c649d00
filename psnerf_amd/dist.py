"""Ray / pixel data parallelism over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).
Every rank holds a replica of all MLP weights (2.7-3.2 MB fp32) and renders a disjoint slice of the
rays / pixels of the SAME view with the FULL light set.  The only exchange per step is
  (1) one tiny all-reduce of the data-dependent loss denominators (mask counts), before backward, and
  (2) ONE flat fp32 bucket all-reduce(SUM) of all dense gradients plus the values of the touched
      light-table rows, after backward.
At ~3 MB the collective is latency-bound on xGMI (7 links x ~153 GB/s per GPU), so a single bucket --
not per-parameter calls, not a bucket sized for NVSwitch -- is the right shape.
Losses are normalised by GLOBAL counts so that sum_r grad_r equals the single-GPU gradient exactly.
"""
import os

import torch
import torch.distributed as dist

PIXEL_KEYS_DIM1 = ('uv', 'object_mask', 'surface_mask', 'points', 'normal', 'vis_train_gt', 'visibility',
                   'sampling_idx')
# (a batch's 'surface_idx' -- the index list of its surface pixels, see PSNetwork.forward -- is rebuilt for the shard)


class DataParallel(object):
    def __init__(self, device=None, force=False):
        """``force``: take the data-parallel code path (bucket views, collectives) even in a world of ONE rank -- the
        collectives are then identities, which is how the RCCL path is exercised on a single-GPU box."""
        up = dist.is_available() and dist.is_initialized()
        self.enabled = up and (dist.get_world_size() > 1 or bool(force))
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self.device = device
        self._buckets = {}
        self.global_rays = None  # ray count of the last stage-1 batch BEFORE sharding (identical on every rank)
        self.n_allreduce = 0
        self.allreduce_bytes = 0

    # ---- sharding -------------------------------------------------------------------------------
    def slice_bounds(self, n):
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per)

    def shard_stage2(self, model_input, ground_truth):
        """Pixel slice of a stage-2 batch (light arrays stay whole)."""
        n = model_input['uv'].shape[1]
        lo, hi = self.slice_bounds(n)
        mi = dict(model_input)
        for k in PIXEL_KEYS_DIM1:
            if k in mi and torch.is_tensor(mi[k]) and mi[k].dim() >= 2 and mi[k].shape[1] == n:
                mi[k] = mi[k][:, lo:hi].contiguous()
        if 'surface_idx' in mi:  # index list of the shard's own surface pixels (one nonzero at sharding time)
            mi['surface_idx'] = mi['surface_mask'][0].nonzero(as_tuple=True)[0]
        gt = dict(ground_truth)
        if 'rgb' in gt:
            gt['rgb'] = gt['rgb'][:, lo:hi].contiguous()
        return mi, gt

    def shard_rays(self, pixels):
        """stage-1: rank slice of the sampled pixel list [1,N,2].  The GLOBAL ray count -- the denominator of the rgb loss
        (losses.py:17-19) -- is remembered here: it is known arithmetically on every rank, no collective needed."""
        self.global_rays = int(pixels.shape[1])
        lo, hi = self.slice_bounds(pixels.shape[1])
        return pixels[:, lo:hi].contiguous()

    def shard_ray_noise(self, noise, n_rays):
        """Rank slice of injected PER-RAY draw tables ('full' [N, S], 'nbr_full' [N, 3]; Renderer._unisurf_sync_free) that
        were made for the whole ray set; group-sized tables (the reference's draw order) are the caller's to slice."""
        if not noise:
            return noise
        lo, hi = self.slice_bounds(n_rays)
        out = dict(noise)
        for k in ('full', 'nbr_full'):
            if out.get(k) is not None and out[k].reshape(-1, out[k].shape[-1]).shape[0] == n_rays:
                out[k] = out[k].reshape(n_rays, -1)[lo:hi].contiguous()
        return out

    # ---- loss denominators ----------------------------------------------------------------------
    def new_step(self):
        pass

    def global_count(self, mask):
        """Number of True elements of ``mask`` summed over ranks (python int).  Every rank must call this the
        same number of times per step (it is a collective): callers compute it once and pass the value on."""
        c = mask.sum().to(torch.int64).reshape(1)
        if self.enabled:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return int(c.item())

    def global_count_tensor(self, mask):
        """Number of True elements of ``mask`` summed over ranks, as a device float tensor [1]: one tiny reduction launch
        (+ one all-reduce under data parallelism) and NO host synchronisation -- the fused loss kernels divide by it on
        the device.  Every rank must call this the same number of times per step."""
        c = mask.sum().to(torch.float32).reshape(1)
        if self.enabled:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return c

    def masked_count_tensor(self, mask_a, mask_b, out=None):
        """global_count_tensor(mask_a & mask_b) with the local count formed by ONE launch (psn_mask_count) on device masks;
        ``out``: a float32 device tensor [1] that receives it (the count buffer of a captured graph)."""
        if mask_a.is_cuda and mask_a.dtype == torch.bool and mask_b.dtype == torch.bool:
            from . import hip
            c = hip.mask_count(mask_a.contiguous(), mask_b.contiguous(), out=out)
            if self.enabled:
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
            return c
        return self.global_count_tensor(mask_a & mask_b)

    def all_reduce_sum_(self, t):
        """Sum a small device tensor over ranks in place (the mask counts of a step in ONE collective); returns it."""
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def global_sum_int(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    # ---- gradients ------------------------------------------------------------------------------
    def _bucket(self, params):
        """The persistent flat fp32 gradient bucket of this parameter set: after allreduce_grads every ``p.grad`` is a VIEW
        into one contiguous buffer (like DDP's gradient_as_bucket_view): one multi-tensor copy in, nothing out."""
        key = tuple(id(p) for p in params)
        b = self._buckets.get(key)
        if b is not None and not all(r() is p for r, p in zip(b[2], params)):
            b = None  # ids are reused once a model is freed: the bucket belongs to THESE parameter objects (weak references)
        if len(self._buckets) > 8:  # train_fix phases x (dense, light tables): a handful; anything more is a leak
            self._buckets = {k: v for k, v in self._buckets.items() if all(r() is not None for r in v[2])}
        if b is None:
            pad = lambda n: (n + 63) // 64 * 64  # views start on 256-byte boundaries, the layout optim.FlatAdam gives the parameters
            total = sum(pad(p.numel()) for p in params)
            flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
            from .optim import register_flat_buffer
            register_flat_buffer(flat)  # its alignment gaps hold zeros: optim.FlatAdam may step its views as merged ranges
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += pad(p.numel())
            import weakref
            b = self._buckets[key] = (flat, views, [weakref.ref(p) for p in params])
        return b

    def prepare_grads(self, params):
        """Replaces ``optimizer.zero_grad()`` under data parallelism: every ``.grad`` is dropped (no launch), so autograd hands
        each parameter its gradient tensor instead of adding it to a zero-filled one (one ``add_`` launch per parameter);
        ``allreduce_grads`` gathers them into the flat bucket with one multi-tensor copy."""
        for p in params:
            p.grad = None  # also for a parameter frozen since the last step: optimisers skip None, like after zero_grad()

    def gather_grads(self, params):
        """The device half in front of the collective: every dense .grad autograd produced is copied into the flat bucket
        (one multi-tensor launch; slots of parameters without a gradient are zeroed) and re-pointed at its bucket view."""
        if not self.enabled:
            return
        params = [p for p in params if p.requires_grad]
        if params:
            flat, views = self._bucket(params)[:2]
            have = [(p, v) for p, v in zip(params, views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
            none = [v for p, v in zip(params, views) if p.grad is None]
            if len(none) == len(params):
                flat.zero_()  # a rank without a graph (an empty pixel slice) contributes zeros
            elif none:
                torch._foreach_zero_(none)
            if have:  # one multi-tensor copy for all parameters (the alignment gaps of the bucket stay zero)
                torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
            for p, v in zip(params, views):
                p.grad = v

    def allreduce_bucket(self, params):
        """ONE all-reduce(SUM) of the flat bucket gather_grads filled; the ``.grad`` views then hold the reduced gradients."""
        if not self.enabled:
            return
        params = [p for p in params if p.requires_grad]
        if params:
            flat = self._bucket(params)[0]
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            self.n_allreduce += 1
            self.allreduce_bytes = flat.numel() * flat.element_size()

    def allreduce_grads(self, params, sparse_params=()):
        """One flat-bucket all-reduce(SUM) over every dense .grad (the tensors autograd produced are first copied into the
        bucket -- one multi-tensor launch --; afterwards every ``.grad`` is its view of the reduced bucket).  Coalesced values of sparse .grad
        (CPU / oracle embeddings; the HIP trainers use dense tables + RowSparseAdam, whose [n_lights, 4] floats ride
        in the bucket whole: 30 KB of a 2.7 MB message, cheaper than four gather / scatter launches) go out as a second
        small message.  A parameter without a gradient on this rank (an empty pixel slice) contributes zeros."""
        if not self.enabled:
            return
        self.gather_grads(params)
        self.allreduce_bucket(params)
        sparse = [(p, p.grad.coalesce()) for p in sparse_params if p.grad is not None]
        if sparse:
            vals = torch.cat([g.values().reshape(-1) for _, g in sparse])
            dist.all_reduce(vals, op=dist.ReduceOp.SUM)
            off = 0
            for p, g in sparse:
                n = g.values().numel()
                p.grad = torch.sparse_coo_tensor(g.indices(), vals[off:off + n].view_as(g.values()), g.shape)
                off += n

    def time_allreduce(self, n_floats, iters=20):
        """Average wall time (ms) of one all-reduce of ``n_floats`` fp32 on this process group (barrier-bracketed)."""
        if not self.enabled:
            return 0.0
        import time
        buf = torch.zeros(int(n_floats), device=self.device)
        for _ in range(3):
            dist.all_reduce(buf)
        if buf.is_cuda:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_reduce(buf)
        if buf.is_cuda:
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def barrier(self):
        if self.enabled:
            dist.barrier()


def init_from_env(backend=None, set_device=True, force=False):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  ``force``: create the
    process group even for a world of one rank (the single-GPU RCCL smoke test; see DataParallel(force=True))."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not force:
        return 0, 0, 1
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl' and set_device:
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
