"""Stage-1 loss with the reference's interface (stage1/model/losses.py:6-70).  Scalar reductions on
[N,3] tensors: host-side torch ops.  ``denoms`` lets the data-parallel trainer substitute GLOBAL
denominators: the reference divides by the batch ray count, the hit count, the
normal-mask count and the valid-mask count."""
import torch
from torch import nn
from torch.nn import functional as F


class Loss(nn.Module):
    def __init__(self, full_weight, grad_weight, norm_weight=1.0, mask_weight=1.0, device=None):
        super().__init__()
        self.full_weight, self.grad_weight = full_weight, grad_weight
        self.norm_weight, self.mask_weight = norm_weight, mask_weight
        self.device = device
        self.global_sum = None  # callable(int) -> int summed over ranks (set by the DP trainer)
        self.global_count = None  # callable(bool mask) -> device float [1], summed over ranks (sync-free path)
        self.global_counts = None  # callable(device float tensor [k]) summing it over ranks IN PLACE (one collective for all masks)
        self.fused = True          # sync-free outputs on the GPU: csrc/loss1.hip (False: the torch formulation below)
        self.global_rays = None   # callable() -> ray count of the batch BEFORE sharding (set by the DP trainer; arithmetic,
                                  # identical on every rank: no collective, so ranks cannot diverge on it -- ADVICE r2)

    def _g(self, n):
        return n if self.global_sum is None else self.global_sum(n)

    def _count(self, mask):
        """Masked-element count as a device tensor (no host synchronisation), summed over ranks under DP."""
        if self.global_count is not None:
            return self.global_count(mask)
        return mask.sum().to(torch.float32).reshape(1)

    def _forward_fused(self, out_dict, rgb_gt, normal_gt, norm_mask, mask, mask_gt, mask_valid):
        """The sync-free branch of forward() as ops.Stage1Losses (two launches, one backward; same terms, same counts)."""
        from .. import ops
        rgb, normal = out_dict['rgb'], out_dict.get('normal_pred')
        n_rays = rgb.shape[1] if self.global_rays is None else self.global_rays()
        with_norm = normal is not None and normal_gt is not None
        with_mask = mask is not None and mask_gt is not None
        flat3 = lambda t: t.reshape(-1, 3)
        loss, terms = ops.Stage1Losses.apply(
            flat3(rgb), flat3(rgb_gt), out_dict['diff_norm_full'].reshape(-1), out_dict['mask_pred'].reshape(-1),
            flat3(normal) if with_norm else None, flat3(normal_gt) if with_norm else None,
            norm_mask.reshape(-1).bool() if with_norm else None, mask.reshape(-1) if with_mask else None,
            mask_gt.reshape(-1) if with_mask else None, mask_valid.reshape(-1).bool() if with_mask else None, int(n_rays),
            (self.full_weight, self.grad_weight, self.norm_weight, self.mask_weight), self.global_counts)
        out = {'fullrgb_loss': terms[0], 'grad_loss': terms[1]}
        if with_norm:
            out['normal_loss'] = terms[2]
        if with_mask:
            out['mask_loss'] = terms[3]
        out['loss'] = loss
        return out

    def forward(self, out_dict, rgb_gt, normal_gt=None, norm_mask=None, mask=None, mask_gt=None, mask_valid=None,
                norm_count=None, valid_count=None):
        """``norm_count`` / ``valid_count``: local counts of ``norm_mask`` / ``mask_valid`` when the caller already has
        them (both masks are inputs: the trainer counts them BEFORE the network call so that no host synchronisation
        sits between forward and backward)."""
        rgb, diff_norm, normal = out_dict['rgb'], out_dict['diff_norm'], out_dict.get('normal_pred')
        dev = rgb.device
        if (self.fused and rgb.is_cuda and out_dict.get('diff_norm_full') is not None and rgb.shape[0] == 1 and norm_count is None
                and valid_count is None and (self.global_count is None or self.global_counts is not None)):
            return self._forward_fused(out_dict, rgb_gt.to(dev), normal_gt, norm_mask, mask, mask_gt, mask_valid)
        zero = torch.zeros((), device=dev)  # a fill kernel: torch.tensor(0.0, device=...) is a pageable H2D copy = a stream sync
        rgb_gt = rgb_gt.to(dev)
        if self.full_weight != 0.0:
            n_rays = rgb.shape[1] if self.global_rays is None else self.global_rays()
            l_rgb = (rgb - rgb_gt).abs().sum() / float(n_rays)
        else:
            l_rgb = zero
        dev_counts = out_dict.get('diff_norm_full') is not None  # the renderer's sync-free training forward
        if dev_counts and self.grad_weight != 0.0:
            # mean over the hit rays of a full-size [N] tensor: masked sum / device-resident hit count (0 for no hits)
            hit = out_dict['mask_pred'].reshape(-1)
            dn = out_dict['diff_norm_full']
            l_grad = torch.where(hit, dn, dn.new_zeros(())).sum() / self._count(hit).clamp(min=1.0)[0]
        elif diff_norm is not None and self.grad_weight != 0.0:
            n_hit = self._g(diff_norm.shape[0])
            l_grad = zero if n_hit == 0 else diff_norm.sum() / float(n_hit)
        else:
            l_grad = zero
        loss = self.full_weight * l_rgb + self.grad_weight * l_grad
        terms = {'fullrgb_loss': l_rgb, 'grad_loss': l_grad}
        if normal is not None and normal_gt is not None and dev_counts and norm_count is None:
            d_n = (normal - normal_gt).abs()
            l_n = torch.where(norm_mask.bool().unsqueeze(-1), d_n, d_n.new_zeros(())).sum() / self._count(norm_mask.bool()).clamp(min=1.0)[0]
            loss = loss + self.norm_weight * l_n
            terms['normal_loss'] = l_n
        elif normal is not None and normal_gt is not None:
            cnt = self._g(int(norm_mask.sum()) if norm_count is None else norm_count)
            if cnt > 0:
                # masked sum instead of the reference's boolean gathers (each a nonzero + host synchronisation)
                # (where, not a product: non-finite values outside the mask are ignored like in the reference)
                d_n = (normal - normal_gt).abs()
                l_n = torch.where(norm_mask.bool().unsqueeze(-1), d_n, d_n.new_zeros(())).sum() / float(cnt)
                loss = loss + self.norm_weight * l_n
                terms['normal_loss'] = l_n
        if mask is not None and mask_gt is not None and dev_counts and valid_count is None:
            bce = F.binary_cross_entropy(mask.clamp(0, 1), mask_gt, reduction='none')
            l_m = torch.where(mask_valid.bool(), bce, bce.new_zeros(())).sum() / self._count(mask_valid.bool()).clamp(min=1.0)[0]
            loss = loss + self.mask_weight * l_m
            terms['mask_loss'] = l_m
        elif mask is not None and mask_gt is not None:
            cnt = self._g(int(mask_valid.sum()) if valid_count is None else valid_count)
            bce = F.binary_cross_entropy(mask.clamp(0, 1), mask_gt, reduction='none')  # log terms are clamped at -100: finite
            l_m = torch.where(mask_valid.bool(), bce, bce.new_zeros(())).sum() / float(max(cnt, 1))
            loss = loss + self.mask_weight * l_m
            terms['mask_loss'] = l_m
        terms['loss'] = loss
        return terms
