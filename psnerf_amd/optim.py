"""Optimiser for the per-light embedding tables of stage 2 (stage2/trainer.py:126-168 uses torch.optim.SparseAdam).

``RowSparseAdam`` has SparseAdam's semantics -- only the rows present in the step's gradient advance their moments and
move; the others stay untouched -- and its state layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter, the same
param_group keys), so optimizer checkpoints of the reference load unchanged.  What differs is the mechanics:
SparseAdam coalesces the sparse gradient (sort + unique + a host synchronisation, 1.2 ms of a 30 ms step) and updates
through sparse tensors; here the touched rows are a 0/1 column and the update is a handful of dense elementwise ops on
the (tiny) table, with no synchronisation.  For the touched rows the arithmetic is SparseAdam's, operation by
operation (torch/optim/sparse_adam.py::_functional.sparse_adam):
    m += (g - m) (1 - b1);  v += (g^2 - v) (1 - b2);  p += -(lr sqrt(1 - b2^t) / (1 - b1^t)) * m / (sqrt(v) + eps)
"""
import math

import torch


class StepScalars(object):
    """The step-dependent scalars of an optimiser launch (bias corrections x learning rate) in DEVICE memory, so that the launch
    itself is identical from step to step and can be replayed from a HIP graph (stage2/graph.py).  ``push(values)`` copies
    this step's values host -> device on the current stream through a ring of pinned rows (a row is rewritten only after
    the copy that last read it has completed: the host may run many steps ahead of the GPU); while the current stream is
    being captured the values are only remembered -- ``flush()`` sends them once the capture has ended, before the replay."""

    def __init__(self, n, device, slots=64):
        self.n = int(n)
        self.dev = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.pin = torch.zeros(slots, self.n, dtype=torch.float32).pin_memory()
        self.events = [None] * slots
        self.turn = 0
        self.pending = None

    def push(self, values):
        values = [float(v) for v in values]
        assert len(values) <= self.n
        if torch.cuda.is_current_stream_capturing():
            self.pending = values
            return
        k = self.turn % len(self.events)
        self.turn += 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        row = self.pin[k]
        row[:len(values)] = torch.tensor(values, dtype=torch.float32)  # (double -> float32: the rounding of a by-value float argument)
        self.dev[:len(values)].copy_(row[:len(values)], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev

    def flush(self):
        if self.pending is not None:
            v, self.pending = self.pending, None
            self.push(v)


class RowSparseAdam(torch.optim.Optimizer):
    graph_scalars = None  # a StepScalars: the fused launch reads its step sizes from device memory (HIP-graph replay)

    def graph_advance(self, plan=None):
        """The host half of one REPLAYED step: the step counts of the tables of the captured launch advance and the step
        sizes of this step go to the device (the device half is the replayed psn_row_adam_dev launch).  ``plan``: the
        (table, group) list the captured launch was built from -- a graph keeps the one of ITS capture (stage2/graph.py);
        the optimiser's own ``_graph_plan`` is that of the LAST fused step of any signature."""
        vals = []
        for p, group in (self._graph_plan if plan is None else plan):
            state = self.state[p]
            state['step'] = int(state['step']) + 1
            t, (b1, b2) = int(state['step']), group['betas']
            vals.append(group['lr'] * math.sqrt(1 - b2 ** t) / (1 - b1 ** t))
        self.graph_scalars.push(vals)

    def _fused_step(self, rows):
        """All tables in ONE launch (psn_row_adam, csrc/loss.hip) when every gradient is a dense fp32 device tensor;
        the torch formulation below is ~14 elementwise launches per table."""
        from . import hip
        items = []
        for group in self.param_groups:
            b1, b2 = group['betas']
            assert not group.get('maximize', False), 'RowSparseAdam: maximize is not supported (param_group edited after construction)'
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    return False
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                t = int(state['step']) + 1
                items.append((p, state, b1, b2, group['eps'], group['lr'] * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)))
        if not items:
            # no table has a gradient (a batch without a surface pixel): nothing to do, in any formulation -- and no launch: a
            # graph captured around this step must not inherit the plan of an earlier fused step (its replays would advance
            # the tables' step counts without an update)
            if self.graph_scalars is not None:
                self._graph_plan = []
            return None
        if len(items) > 4:
            return False
        for p, state, *_ in items:
            state['step'] = int(state['step']) + 1
        dev = None
        if self.graph_scalars is not None:
            self.graph_scalars.push([it[-1] for it in items])
            self._graph_plan = [(p, next(g for g in self.param_groups if any(q is p for q in g['params']))) for p, *_ in items]
            dev = self.graph_scalars.dev
        hip.row_adam([(p, p.grad, st['exp_avg'], st['exp_avg_sq'], b1, b2, eps, ss) for p, st, b1, b2, eps, ss in items],
                     rows.contiguous(), step_sizes_dev=dev)
        return True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, maximize=False):
        if not 0.0 < lr:
            raise ValueError('Invalid learning rate: %r' % (lr,))
        if maximize:
            raise ValueError('maximize is not supported')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, maximize=maximize))

    @torch.no_grad()
    def step(self, rows=None):
        """rows: 1-D index tensor of the table rows used by this step (duplicates allowed).  Required when the
        gradients are dense (nn.Embedding(sparse=False)); with sparse gradients the rows are their indices."""
        if rows is not None and rows.is_cuda:
            fused = self._fused_step(rows)
            if fused or fused is None:
                return
        if not any(p.grad is not None for group in self.param_groups for p in group['params']):
            return
        # (an EAGER step may take the torch formulation beside live graphs -- its step sizes are host scalars; a capture may not)
        assert not (self.graph_scalars is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()), \
            'RowSparseAdam: only the fused launch can be captured into a graph'

        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse:  # uncoalesced is fine: duplicates are summed by index_add_
                    idx = g._indices()[0]
                    g = torch.zeros_like(p).index_add_(0, idx, g._values())
                else:
                    if rows is None:
                        raise RuntimeError('RowSparseAdam.step: dense gradients need the rows of this step')
                    idx = rows
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                state['step'] += 1
                t = int(state['step'])
                touched = torch.zeros(p.shape[0], *([1] * (p.dim() - 1)), dtype=p.dtype, device=p.device).index_fill_(0, idx, 1.0)
                m, v = state['exp_avg'], state['exp_avg_sq']
                m.add_(g.sub(m).mul_(1 - beta1).mul_(touched))
                v.add_(g.pow(2).sub_(v).mul_(1 - beta2).mul_(touched))
                step_size = group['lr'] * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
                p.add_(m.div(v.sqrt().add_(group['eps'])).mul_(-step_size).mul_(touched))


def _pad64(n):
    return (int(n) + 63) // 64 * 64


# Flat gradient allocations whose alignment gaps are KNOWN to hold zeros: a FlatAdam's own gradient buffer and the
# data-parallel buckets (dist.DataParallel._bucket registers them).  Only views of these may be stepped as ranges that
# merge across a gap; a gradient that happens to be a view of some other 1-D tensor is copied like any loose gradient.
_FLAT_BUFFERS = []


def register_flat_buffer(t):
    import weakref
    _FLAT_BUFFERS[:] = [r for r in _FLAT_BUFFERS if r() is not None]
    if not any(r() is t for r in _FLAT_BUFFERS):
        _FLAT_BUFFERS.append(weakref.ref(t))


def _known_flat_base(g):
    """The registered flat buffer ``g`` is a contiguous view of, or None."""
    base = g._base if (torch.is_tensor(g) and not g.is_sparse) else None
    if base is None or base.dim() != 1 or base.dtype != torch.float32 or not base.is_contiguous() or not g.is_contiguous():
        return None
    for r in _FLAT_BUFFERS:
        t = r()
        if t is not None and t.data_ptr() == base.data_ptr() and t.numel() == base.numel():
            return base
    return None


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad off, no weight decay: what stage2/trainer.py:126-133 constructs) over ONE flat fp32 buffer.

    Parameters and moments are contiguous allocations; the gradients autograd produced are gathered into a third one by one
    multi-tensor copy at the start of ``step()`` (under data parallelism they already are views of the all-reduced bucket),
    so a step is one launch of psn_adam_flat with torch's arithmetic (seven foreach launches there).  ``param.data`` of every parameter is re-pointed into the flat buffer at
    construction (values preserved; in-place loads such as load_state_dict keep working, and a parameter whose storage was
    replaced from outside is detected and re-attached).  State layout and keys are torch.optim.Adam's (``step``,
    ``exp_avg``, ``exp_avg_sq`` per parameter; the same param_group keys), so optimiser checkpoints are interchangeable;
    parameters without a gradient are skipped, and parameters that start training later have their own step count.
    CPU / non-fp32 parameters fall back to the same update written with torch ops."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr:
            raise ValueError('Invalid learning rate: %r' % (lr,))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None))
        self._flat = None
        self.graph_scalars = None  # a StepScalars: the launches read (neg_step, bc2_sqrt) of every range from device memory
        self._graph_plan = None

    def graph_advance(self, plan=None):
        """The host half of one REPLAYED step: the step counts of the parameters of the captured launches advance and this
        step's scalars go to the device (the device half are the replayed gather + psn_adam_flat_dev launches).  ``plan``:
        the (group, members, range representatives) list of the capture that is replayed (stage2/graph.py snapshots it per
        graph; ``_graph_plan`` itself is overwritten by every step of any other signature)."""
        import math
        vals = []
        for group, params, reps in (self._graph_plan if plan is None else plan):
            b1, b2 = group['betas']
            for p in params:
                self.state[p]['step'] += 1
            for rep in reps:
                t = int(self.state[rep]['step'])
                vals += [-(group['lr'] / (1 - b1 ** t)), math.sqrt(1 - b2 ** t)]
        self.graph_scalars.push(vals)

    # ---- flat storage -----------------------------------------------------------------------------------------------
    def _all_params(self):
        return [p for g in self.param_groups for p in g['params']]

    def _attach(self):
        ps = self._all_params()
        ok = all(p.is_cuda and p.dtype == torch.float32 for p in ps) and len(ps) > 0
        if not ok:
            self._flat = False
            return
        # every parameter starts on a 256-byte boundary, like an allocation of its own would (kernels read weights with 16-byte
        # loads); the gaps hold zeros in all four buffers, and Adam maps zeros to zeros there
        total = sum(_pad64(p.numel()) for p in ps)
        dev = ps[0].device
        old = self._flat if isinstance(self._flat, dict) else None
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        m, v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        offs, off = {}, 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                st = self.state.get(p, {})
                for key, buf in (('exp_avg', m), ('exp_avg_sq', v)):
                    if key in st:
                        buf[off:off + n].copy_(st[key].reshape(-1))
                        st[key] = buf[off:off + n].view_as(p)
                p.data = flat[off:off + n].view_as(p)
                offs[p] = off
                off += _pad64(n)
        self._flat = dict(p=flat, m=m, v=v, off=offs)
        del old

    def _attached(self):
        if self._flat is None:
            self._attach()
        if not self._flat:
            return False
        f = self._flat
        base = f['p'].data_ptr()
        if any(p.data_ptr() != base + 4 * f['off'][p] for p in self._all_params()):
            self._attach()  # a storage was replaced from outside (.to(), assign-style load): take the values over again
        return bool(self._flat)

    def attach_grads(self):
        """Replaces ``zero_grad()``: every ``.grad`` is dropped (no launch); autograd then HANDS each parameter its gradient
        instead of adding it to a zero-filled view (one ``add_`` launch per parameter), and ``step()`` gathers the
        gradients into the flat buffer with one multi-tensor copy.  Returns whether the parameters are flat."""
        self.zero_grad(set_to_none=True)
        return self._attached()

    def _gather_grads(self):
        """Copies every dense fp32 gradient that is not yet a view of a flat allocation into this optimiser's flat gradient
        buffer (laid out like the parameters; alignment gaps stay zero) and re-points ``.grad`` at the views."""
        f = self._flat
        loose = []
        for p in self._all_params():
            g = p.grad
            if g is None or g.is_sparse or g.dtype != torch.float32 or not g.is_cuda:
                continue
            if g._base is not None and _known_flat_base(g) is not None:
                continue  # already a view of a flat buffer (this one, or the data-parallel bucket)
            loose.append(p)  # (a tensor autograd handed over has no base: no registry lookup)
        if not loose:
            return
        if f.get('g') is None:
            f['g'] = torch.zeros_like(f['p'])
            f['gview'] = {}
            register_flat_buffer(f['g'])
        gv = f['gview']
        views = []
        for p in loose:  # the view of a parameter's slot is made once, not once per step
            v = gv.get(p)
            if v is None:
                v = gv[p] = f['g'][f['off'][p]:f['off'][p] + p.numel()].view_as(p)
            views.append(v)
        torch._foreach_copy_(views, [p.grad for p in loose])
        for p, v in zip(loose, views):
            p.grad = v

    # ---- checkpoints: never serialise views of the whole flat allocation ----------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        # torch returns the LIVE per-parameter dictionaries (sd['state'][i] is self.state[p]): build fresh ones, never
        # write into them -- replacing the moment views by clones would freeze every later checkpoint at this one
        sd['state'] = {k: {kk: (t.detach().clone() if torch.is_tensor(t) else t) for kk, t in st.items()}
                       for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None  # moments were replaced by loaded copies: re-attach (values are taken over) at the next step

    # ---- the update ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self):
        import math
        from . import hip
        flat_ok = self._attached()
        if flat_ok:
            self._gather_grads()
        plan, launches = [], []
        for group in self.param_groups:
            assert not group.get('maximize', False) and not group.get('amsgrad', False) and group.get('weight_decay', 0) == 0, \
                'FlatAdam: plain Adam only'
            b1, b2 = group['betas']
            lr, eps = group['lr'], group['eps']
            segs, gbuf, members, reps = [], None, [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    if flat_ok:
                        f, o, n = self._flat, self._flat['off'][p], p.numel()
                        st['exp_avg'], st['exp_avg_sq'] = f['m'][o:o + n].view_as(p), f['v'][o:o + n].view_as(p)
                        st['exp_avg'].zero_()
                        st['exp_avg_sq'].zero_()
                    else:
                        st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
                st['step'] += 1
                t = int(st['step'])
                neg_step, bc2s = -(lr / (1 - b1 ** t)), math.sqrt(1 - b2 ** t)
                g = p.grad
                if flat_ok and self._flat.get('gview') is not None and self._flat['gview'].get(p) is g:
                    base = self._flat['g']  # this optimiser's own slot view (set by _gather_grads): no registry lookup
                else:
                    base = _known_flat_base(g) if flat_ok else None
                if base is not None and (gbuf is None or (base.data_ptr() == gbuf.data_ptr() and base.numel() == gbuf.numel())):
                    # the gradient is a view of ONE flat allocation (attach_grads, or the data-parallel bucket): a range of the launch
                    gbuf = base
                    o, go, n = self._flat['off'][p], (g.data_ptr() - base.data_ptr()) // 4, p.numel()
                    gap = o - (segs[-1][0] + segs[-1][2]) if segs else -1
                    members.append(p)
                    if segs and 0 <= gap < 64 and gap == go - (segs[-1][1] + segs[-1][2]) and segs[-1][3:] == (neg_step, bc2s):
                        # the next parameter of both layouts, behind the same (zero-filled) alignment gap: one range
                        segs[-1] = (segs[-1][0], segs[-1][1], segs[-1][2] + gap + n, neg_step, bc2s)
                    else:
                        segs.append((o, go, n, neg_step, bc2s))
                        reps.append(p)
                    continue
                assert not (self.graph_scalars is not None and torch.cuda.is_current_stream_capturing()), \
                    'FlatAdam: a gradient outside the flat layout cannot be captured into a graph'
                # torch's formulation for whatever is not in the flat layout
                m, v = st['exp_avg'], st['exp_avg_sq']
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                p.addcdiv_(m, (v.sqrt() / bc2s).add_(eps), value=neg_step)
            if segs:
                launches.append((gbuf, segs, b1, b2, eps))
                plan.append((group, members, reps))
        dev, off = None, 0
        if self.graph_scalars is not None:
            # every launch reads its ranges' scalars from device memory: identical launches from step to step (graph replay)
            self.graph_scalars.push([x for _, segs, *_ in launches for s_ in segs for x in s_[3:]])
            self._graph_plan = plan
            dev = self.graph_scalars.dev
        for gbuf, segs, b1, b2, eps in launches:
            f = self._flat
            hip.adam_flat(f['p'], gbuf, f['m'], f['v'], segs, b1, b2, eps, scalars_dev=None if dev is None else dev[off:])
            off += 2 * len(segs)
