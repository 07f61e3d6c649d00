"""The stage-2 train step replayed from HIP graphs (the body of stage2/trainer.py:355-410 for a per-rank batch that is too
small to hide the host: BASELINE cfg 4 = cfg 3's 32768 pixels sharded over 8 ranks = 4096 pixels per rank, where the
~110 launches of a step cost more host time than their kernels take).

``GraphedTrainStep(step)`` wraps a ``TrainStep``; every ``.step(...)`` call is exactly one optimisation step with the
eager step's arithmetic, bit for bit:

  * the first ``warmup`` calls of a new signature (input shapes, trainable set, loss weights -- i.e. per train_fix phase
    and batch geometry) run ``TrainStep.step`` eagerly (on the capture stream, so every cache -- workspaces, weight packs,
    optimiser state, the data-parallel bucket -- exists before the capture);
  * the next call CAPTURES the device work of a step (a capture records, it does not execute) and replays it;
  * every further call copies the batch into the captured input buffers (skipped for tensors that already ARE those
    buffers: a data pipeline can write into ``.buffers`` directly) and replays.

ALIASING: a replayed step returns the graph's STATIC output tensors (``terms``, ``out``): the next replay of the same
signature overwrites them in place (the eager step returns fresh tensors).  Read what you need (``float(t)``, ``.clone()``)
before the next ``.step``; ``GraphedTrainStep(..., clone_terms=True)`` returns detached clones of the scalar loss terms
(one multi-tensor copy per step) for callers that keep them across steps.

What stays on the host per replayed step: train_fix, the iteration / scheduler bookkeeping, the optimisers' step counts and
their step-dependent scalars (bias corrections x learning rate), which the Adam / SparseAdam launches read from DEVICE memory
(``optim.StepScalars``, psn_adam_flat_dev / psn_row_adam_dev) so that the captured launches never change; the vis_plus draw
(np.random on the host + two gathers); and under data parallelism the two collectives, which are NOT captured: the step is then
two graphs, [forward, losses, backward, gradients gathered into the flat bucket] -> RCCL all-reduce of the bucket ->
[Adam, SparseAdam], with the 4-byte count all-reduce in front.

Measured on this ROCm (7.x): a single-stream graph costs the host 0.04 ms per replay, a graph that forks to the model's side
stream 2.5 - 3 ms (still below the step's GPU time); kernel-to-kernel gaps inside a replay equal those of a host that runs
ahead -- a HIP graph removes the HOST from the critical path, not GPU time.
"""
import torch

from .. import hip
from ..optim import FlatAdam, RowSparseAdam, StepScalars

_INPUT_KEYS_SKIP = ()


class _Captured(object):
    def __init__(self):
        self.graphs = []
        self.inp = self.gt = self.l_slt = self.noise = self.count = None
        self.terms = self.out = None
        self.trainable = None
        self.train_light = False
        self.sg_plan = self.light_plan = None  # the optimisers' host-side plans of this capture (optim.*.graph_advance)


class GraphedTrainStep(object):
    SKIP_PADDING = True  # pad_to_pixels: hand the device-side surface count to the model (False: the padding rows are evaluated; A/B)

    def __init__(self, step, warmup=2, max_graphs=8, overlap_small_nets=None, adopt_inputs=False, agree=None, pad_to_pixels=False,
                 pad_multiple=0, clone_terms=False):
        """``overlap_small_nets``: None keeps the model's setting; False captures a single-stream graph (0.04 ms of host
        time per replay instead of ~2.5 ms, at the price of the side-stream overlap).  ``adopt_inputs``: the tensors of the
        batch that is captured BECOME the graph's input buffers (no clones): for a caller that keeps refilling the same
        tensors -- a data pipeline with fixed staging buffers, the benchmark's resident batch -- a replay then copies
        nothing (a cloned batch costs ~17 device-to-device copies of 5 us per step).  ``agree``: callable(bool) -> bool, called
        right after every capture ATTEMPT with this rank's success and returning the job's verdict (data-parallel jobs pass an
        all-reduce(MIN)): a rank whose capture failed must not leave the others waiting in the replay's collectives -- either
        every rank replays or every rank raises.  ``pad_to_pixels``: the surface-pixel list of every batch is built ON THE DEVICE
        with the fixed length N (psn_surface_index: the real entries, then the last one repeated), so that batches whose surface
        count differs share ONE graph and the step has no host synchronisation even for the reference's dictionary; the dead rows
        behind the real ones contribute exact zeros (their dense outputs are never written, their gradients are zeroed by
        psn_gather_rows_valid); the gradient-free shading rows among them -- 92 % of the rows of the BEAR step -- are not even
        evaluated (the visibility launch reads the count on the device, psn_mlp_infer_padded), the others are: (N - Ns) / Ns
        more rows in the small networks, the supervision rows, shading and losses, and split-K sums in a different order than
        the unpadded step (equal to rounding, not bit for bit).  An EMPTY mask gives no pixel a row (the pixel -> row map reads the
        count too): dense outputs, losses and gradients are those of the eager step on the empty batch.  ``pad_multiple`` = k > 0: for batches that bring their 'surface_idx'
        along (handoff.ViewSampler, the benchmark: the count is then known on the HOST), the list is padded to the next
        multiple of k instead -- at most k - 1 dead rows, one graph per capacity that occurs (``max_graphs``).  With either padding an
        injected ``noise`` (tests; training draws it on the device) has the PADDED row count."""
        assert isinstance(step.sg_optimizer, FlatAdam) and isinstance(step.light_optimizer, RowSparseAdam) and step.FUSED_LOSSES, \
            'GraphedTrainStep needs the device-resident step (FlatAdam, RowSparseAdam, fused losses)'
        self.step_obj, self.warmup, self.max_graphs = step, int(warmup), int(max_graphs)
        self.stream = torch.cuda.Stream(device=step.device)
        self._seen = {}
        self._captured = {}
        if overlap_small_nets is not None:
            step.model.overlap_small_nets = bool(overlap_small_nets)
        self.adopt_inputs = bool(adopt_inputs)
        self.agree = agree
        self.pad_to_pixels = bool(pad_to_pixels)
        self.pad_multiple = int(pad_multiple)
        assert not (self.pad_to_pixels and self.pad_multiple), 'pad_to_pixels and pad_multiple exclude each other'
        # (a data-parallel job captures on every rank at the same step -- `agree` is a collective; capacities that depend on the
        #  rank's own surface count would let one rank capture while another replays: such jobs pad to the pixel count)
        assert not (self.pad_multiple and agree is not None), 'pad_multiple is for single-process jobs; data-parallel jobs use pad_to_pixels'
        self.n_replays = self.n_eager = self.n_captures = 0
        self.clone_terms = bool(clone_terms)

    def reset(self):
        """Drop every captured graph (the next steps of each signature run eagerly and are captured again).  Call it after anything
        that REPLACES tensors the captured launches address -- ``checkpoints.load_stage2`` (optimiser state is re-created by
        ``load_state_dict``), ``model.to(...)``, a new optimiser: a graph holds raw addresses, not tensors."""
        self._captured.clear()
        self._seen.clear()
        for opt in (self.step_obj.sg_optimizer, self.step_obj.light_optimizer):
            if hasattr(opt, '_graph_plan'):
                opt._graph_plan = None

    # ---- signature of a step -------------------------------------------------------------------------------------------
    def _key(self, model_input, ground_truth, l_slt, noise):
        st = self.step_obj
        shapes = tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in model_input.items() if torch.is_tensor(v)))
        gshapes = tuple(sorted((k, tuple(v.shape)) for k, v in ground_truth.items() if torch.is_tensor(v)))
        nshapes = tuple(sorted((k, tuple(v.shape)) for k, v in (noise or {}).items() if torch.is_tensor(v)))
        flags = tuple(p.requires_grad for p in st.model.parameters()) + (st.light_para.weight.requires_grad,
                                                                         st.light_inten_para.weight.requires_grad)
        w = (st.loss.sg_rgb_weight, st.loss.albedo_smooth_weight, st.loss.rough_smooth_weight, st.loss.vis_weight,
             st.loss_n.normal_weight, st.loss_n.normal_smooth_weight, st.model.training)
        return (shapes, gshapes, tuple(l_slt.shape), nshapes, flags, w)

    @property
    def buffers(self):
        """{signature: (model_input, ground_truth, l_slt, noise)} static input tensors of the captured graphs."""
        return {k: (c.inp, c.gt, c.l_slt, c.noise) for k, c in self._captured.items()}

    # ---- one optimisation step -----------------------------------------------------------------------------------------
    def step(self, model_input, ground_truth, l_slt, train_order=True, noise=None, vidx=None):
        st = self.step_obj
        if train_order:
            st.train_fix()
        st.dp.new_step()
        model_input = st.select_vis_lights(model_input, vidx)
        if self.pad_to_pixels:
            model_input = dict(model_input)
            sm = model_input['surface_mask'][0].contiguous()
            # (the count stays on the device: the visibility launch skips the shading rows of the padding with it)
            model_input['surface_idx'], cnt = hip.surface_index(sm, sm.numel())
            if self.SKIP_PADDING:
                model_input['surface_count'] = cnt
        elif self.pad_multiple and model_input.get('surface_idx') is not None and model_input['surface_idx'].numel() > 0:
            idx = model_input['surface_idx']
            ns, k = idx.numel(), self.pad_multiple
            cap = -(-ns // k) * k
            if cap != ns:
                model_input = dict(model_input)
                model_input['surface_idx'] = torch.cat([idx, idx[-1:].expand(cap - ns)])
                if self.SKIP_PADDING:
                    model_input['surface_count'] = torch.full((1,), float(ns), device=idx.device, dtype=torch.float32)
        if 'surface_idx' not in model_input:
            # the reference's dictionary (no index list of the surface pixels): built here, OUTSIDE the graph -- nonzero() is a
            # host synchronisation and cannot be captured; its length is part of the graph's signature
            model_input = dict(model_input)
            model_input['surface_idx'] = model_input['surface_mask'][0].nonzero(as_tuple=True)[0]
        key = self._key(model_input, ground_truth, l_slt, noise)
        cap = self._captured.get(key)
        if cap is None:
            seen = self._seen.get(key, 0)
            if seen < self.warmup or hip.PROFILE_EVENTS is not None:
                self._seen[key] = seen + 1
                return self._eager(model_input, ground_truth, l_slt, noise)
            if self.agree is None:
                cap = self._capture(key, model_input, ground_truth, l_slt, noise)
            else:
                cap, err = None, None
                try:
                    cap = self._capture(key, model_input, ground_truth, l_slt, noise)
                except Exception as e:  # noqa: BLE001  (reported below, on every rank at the same point of the collective sequence)
                    err = e
                if not self.agree(err is None):
                    self._captured.pop(key, None)
                    raise RuntimeError('GraphedTrainStep: the capture failed on %s rank: %r' % ('this' if err is not None else 'another', err))
            first = True
        else:
            first = False
            self._load(cap, model_input, ground_truth, l_slt, noise)
            # the plans of THIS graph's capture: eager steps / captures of other signatures in between overwrite the
            # optimisers' own ``_graph_plan`` (another gradient set, another range layout -- or none at all for a batch
            # without a surface pixel), and a replay must advance the step counts its launches were built from
            st.sg_optimizer.graph_advance(cap.sg_plan)
            if cap.train_light:
                st.light_optimizer.graph_advance(cap.light_plan)
        self._replay(cap)
        if not first:
            st.model.invalidate_packs(trainable_only=True)  # (the captured step re-packs; an eager step that follows must too)
        st._advance(cap.train_light)
        self.n_replays += 1
        if self.clone_terms:
            keys = [k for k, v in cap.terms.items() if torch.is_tensor(v)]
            fresh = torch._foreach_add([cap.terms[k].detach() for k in keys], 0.0)  # one multi-tensor launch
            terms = dict(cap.terms)
            terms.update(zip(keys, fresh))
            return terms, cap.out
        return cap.terms, cap.out

    def _eager(self, model_input, ground_truth, l_slt, noise):
        """A plain TrainStep step on the capture stream (so that every per-stream cache the capture will use exists)."""
        st = self.step_obj
        cur = torch.cuda.current_stream(st.device)
        # The collectives of a data-parallel step are issued on the CALLER's stream, never on the capture stream: RCCL's work
        # objects hold events recorded on the stream they were issued from, its watchdog thread polls them for ~100 ms
        # afterwards, and HIP refuses a query of an event whose stream is capturing by then ("operation not permitted on an
        # event last recorded in a capturing stream": the exception kills the process from the watchdog thread -- seen once
        # in bench.py when the capture followed the warm-up steps quickly enough).
        count = None
        if st.dp.enabled:
            count = st.dp.masked_count_tensor(model_input['surface_mask'], model_input['object_mask'])
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            terms, out, trainable, train_light = st._fwd_bwd(model_input, ground_truth, l_slt, noise=noise, count=count)
        cur.wait_stream(self.stream)
        st._reduce(trainable)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            st._optimise(l_slt, trainable, train_light)
        cur.wait_stream(self.stream)
        for t in list(terms.values()) + list(out.values()):
            if torch.is_tensor(t):
                t.record_stream(cur)
        st._advance(train_light)
        self.n_eager += 1
        return terms, out

    # ---- capture ---------------------------------------------------------------------------------------------------------
    def _capture(self, key, model_input, ground_truth, l_slt, noise):
        st = self.step_obj
        if len(self._captured) >= self.max_graphs:  # batch geometries that keep changing: drop the oldest graph
            self._captured.pop(next(iter(self._captured)))
        cap = _Captured()
        dev = st.device
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            keep = (lambda v: v.detach()) if self.adopt_inputs else (lambda v: v.detach().clone())
            cap.inp = {k: (keep(v) if torch.is_tensor(v) else v) for k, v in model_input.items()}
            cap.gt = {k: (keep(v) if torch.is_tensor(v) else v) for k, v in ground_truth.items()}
            cap.l_slt = keep(l_slt)
            cap.noise = None if not noise else {k: (keep(v) if torch.is_tensor(v) else v) for k, v in noise.items()}
            if st.dp.enabled:
                cap.count = torch.zeros(1, device=dev, dtype=torch.float32)
            for opt, n in ((st.sg_optimizer, 2 * hip.ADAM_MAX_SEGS * 4), (st.light_optimizer, 8)):
                if opt.graph_scalars is None:
                    opt.graph_scalars = StepScalars(n, dev)
        torch.cuda.synchronize(dev)
        # capture_error_mode 'thread_local': other threads of the process (RCCL's watchdog and proxy threads, pinned-memory
        # loaders) keep calling HIP while this thread captures; in the default 'global' mode any such call invalidates the capture
        g_a = torch.cuda.CUDAGraph()
        hip.capturing(True)  # (scratch buffers the captured launches use stay alive when a later, larger batch replaces them)
        try:
            with torch.cuda.graph(g_a, stream=self.stream, capture_error_mode='thread_local'):
                cap.terms, cap.out, cap.trainable, cap.train_light = st._fwd_bwd(cap.inp, cap.gt, cap.l_slt, noise=cap.noise, count=cap.count)
                if not st.dp.enabled:
                    st._optimise(cap.l_slt, cap.trainable, cap.train_light)
            cap.graphs.append(g_a)
            if st.dp.enabled:
                g_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_b, stream=self.stream, pool=g_a.pool(), capture_error_mode='thread_local'):
                    st._optimise(cap.l_slt, cap.trainable, cap.train_light)
                cap.graphs.append(g_b)
        finally:
            hip.capturing(False)
        # the optimisers' host halves ran inside the captures (step counts advanced, this step's scalars remembered): send the
        # scalars now, in front of the replay that executes the captured step
        st.sg_optimizer.graph_scalars.flush()
        st.light_optimizer.graph_scalars.flush()
        cap.sg_plan = list(st.sg_optimizer._graph_plan or [])
        cap.light_plan = list(st.light_optimizer._graph_plan or []) if cap.train_light else []
        self._captured[key] = cap
        self.n_captures += 1
        return cap

    # ---- replay ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _pairs(dst, src, out):
        """(buffer, tensor) pairs of a dictionary that need a copy (a tensor that already IS its buffer does not)."""
        for k, v in src.items():
            d = dst.get(k)
            if torch.is_tensor(v) and torch.is_tensor(d) and d.data_ptr() != v.data_ptr():
                out.append((d, v))

    def _load(self, cap, model_input, ground_truth, l_slt, noise):
        """The batch into the graph's input buffers: ONE launch for every contiguous device tensor of matching type
        (psn_copy_bytes_group; a copy_ each was ~17 launches of 5 us), copy_ for whatever is left (host tensors, other dtypes)."""
        pairs = []
        self._pairs(cap.inp, model_input, pairs)
        self._pairs(cap.gt, ground_truth, pairs)
        if cap.l_slt.data_ptr() != l_slt.data_ptr():
            pairs.append((cap.l_slt, l_slt))
        if cap.noise:
            self._pairs(cap.noise, noise or {}, pairs)
        fast = [(d, v) for d, v in pairs if v.is_cuda and v.dtype == d.dtype and v.numel() == d.numel() and v.is_contiguous() and d.is_contiguous() and v.numel() > 0]
        if len(fast) > 1:
            hip.copy_group(fast)
            done = set(id(d) for d, _ in fast)
            pairs = [(d, v) for d, v in pairs if id(d) not in done]
        for d, v in pairs:
            d.copy_(v, non_blocking=True)

    def _replay(self, cap):
        st = self.step_obj
        if st.dp.enabled:
            # the masked-pixel count of the global batch: formed and all-reduced outside the graph (a collective is not captured)
            st.dp.masked_count_tensor(cap.inp['surface_mask'], cap.inp['object_mask'], out=cap.count)  # counted and all-reduced in place
            cap.graphs[0].replay()
            st.dp.allreduce_bucket(cap.trainable)
            cap.graphs[1].replay()
        else:
            cap.graphs[0].replay()
