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


class RowSparseAdam(torch.optim.Optimizer):
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
        if not items or len(items) > 4:
            return False
        for p, state, *_ in items:
            state['step'] = int(state['step']) + 1
        hip.row_adam([(p, p.grad, st['exp_avg'], st['exp_avg_sq'], b1, b2, eps, ss) for p, st, b1, b2, eps, ss in items],
                     rows.contiguous())
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
        if rows is not None and rows.is_cuda and self._fused_step(rows):
            return
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
