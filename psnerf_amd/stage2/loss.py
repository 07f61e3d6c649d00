"""Stage-2 losses with the reference's interface (stage2/model/loss.py:6-141), device-agnostic
(the reference hard-codes .cuda()).  Small reductions over dense [L,N,3] tensors -- host-side torch ops.

Under ray/pixel data parallelism the masked means use GLOBAL pixel counts (``global_count``) so that the
summed rank gradients equal the single-GPU gradient."""
import torch
from torch import nn
from torch.nn import functional as F


def _masked_mean(diff, mask_b, count, channels):
    """mean over the masked elements of ``diff`` [B,N,C] (mask_b [B,N] bool, ``count`` masked pixels per batch row
    summed over rows) without a boolean gather: sum(where(mask, diff, 0)) / (count * C).  ``where`` rather than a
    multiplication: like the reference's boolean indexing it ignores non-finite values outside the mask (NaN * 0 = NaN)."""
    return torch.where(mask_b.bool().unsqueeze(-1), diff, diff.new_zeros(())).sum() / float(count * channels)


class MainLoss(nn.Module):
    """stage2/model/loss.py:6-92.  The reference selects the masked pixels with boolean indexing (a nonzero +
    host sync per term on a GPU) and takes means; here every term is sum(|a-b| * mask) / count with ONE mask
    count per step.  Under data parallelism ``global_count`` supplies the count summed over ranks."""

    def __init__(self, sg_rgb_weight, loss_type='L1', albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=1.0):
        super().__init__()
        self.sg_rgb_weight = sg_rgb_weight
        self.albedo_smooth_weight = albedo_smooth_weight
        self.rough_smooth_weight = rough_smooth_weight
        self.vis_weight = vis_weight
        if loss_type not in ('L1', 'L2'):
            raise Exception('Unknown loss_type!')
        self.loss_type = loss_type
        self.global_count = None  # set by the data-parallel trainer

    def _img(self, a, b):
        return (a - b).abs() if self.loss_type == 'L1' else (a - b) ** 2

    def forward(self, model_outputs, ground_truth, model_input=None, count=None):
        """``count``: number of pixels in (network_object_mask & object_mask), summed over ranks under data
        parallelism; computed here (one host sync) when the caller does not supply it."""
        m, om = model_outputs['network_object_mask'], model_outputs['object_mask']
        mask = m & om  # [1,N]
        dev = mask.device
        if count is None:
            count = int(mask.sum()) if self.global_count is None else self.global_count(mask)
        zero = torch.zeros((), device=dev)  # a fill kernel: torch.tensor(0.0, device=...) is a pageable H2D copy = a stream sync
        rgb_gt = ground_truth['rgb'].to(dev)
        L = rgb_gt.shape[0]
        if count == 0:
            rgb_loss = zero
        else:
            rgb_loss = _masked_mean(self._img(model_outputs['sg_rgb_values'], rgb_gt), mask.expand(L, -1), count * L, 3)
        loss = self.sg_rgb_weight * rgb_loss
        a_loss = r_loss = None
        if 'albedo_jitter' in model_outputs and self.albedo_smooth_weight > 0:
            x, xj = model_outputs['albedo_values'], model_outputs['albedo_jitter']
            a_loss = zero if count == 0 else _masked_mean((x - xj).abs(), mask.expand(x.shape[0], -1),
                                                          count * x.shape[0], x.shape[-1])
            loss = loss + self.albedo_smooth_weight * a_loss
        if 'rough_jitter' in model_outputs and self.rough_smooth_weight > 0:
            x, xj = model_outputs['rough_values'], model_outputs['rough_jitter']
            r_loss = zero if count == 0 else _masked_mean((x - xj).abs(), mask.expand(x.shape[0], -1),
                                                          count * x.shape[0], x.shape[-1])
            loss = loss + self.rough_smooth_weight * r_loss
        terms = {'sg_rgb_loss': rgb_loss, 'albedo_smooth_loss': a_loss, 'rough_smooth_loss': r_loss}
        if 'visibility' in model_input and 'visibility' in model_outputs:
            if 'vis_train_gt' in model_input and 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['vis_train_gt']
            elif 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['visibility']
            else:
                v, gt = model_outputs['visibility'][..., 0], model_input['visibility']
            if count == 0:
                vis_loss = zero
            else:
                vis_loss = _masked_mean(self._img(v, gt).unsqueeze(-1), mask.expand(gt.shape[0], -1), count * gt.shape[0], 1)
            loss = loss + self.vis_weight * vis_loss
            terms['vis_loss'] = vis_loss
        terms['loss'] = loss
        return terms


class NormalLoss(nn.Module):
    """stage2/model/loss.py:96-141 (same masked-sum formulation)."""

    def __init__(self, normal_weight, normal_smooth_weight=0):
        super().__init__()
        self.normal_weight, self.normal_smooth_weight = normal_weight, normal_smooth_weight
        self.global_count = None

    def forward(self, model_outputs, count=None):
        gt = F.normalize(model_outputs['normal_values'], dim=-1)
        mask = model_outputs['network_object_mask'] & model_outputs['object_mask']
        dev = mask.device
        if count is None:
            count = int(mask.sum()) if self.global_count is None else self.global_count(mask)
        zero = torch.zeros((), device=dev)  # a fill kernel: torch.tensor(0.0, device=...) is a pageable H2D copy = a stream sync
        if count == 0:
            n_loss = zero
        else:
            n_loss = _masked_mean((model_outputs['normal_pred'] - gt) ** 2, mask, count, 3)
        loss = self.normal_weight * n_loss
        s_loss = None
        if 'normal_jitter' in model_outputs and self.normal_smooth_weight > 0:
            s_loss = zero if count == 0 else _masked_mean(
                (model_outputs['normal_pred'] - model_outputs['normal_jitter']).abs(), mask, count, 3)
            loss = loss + self.normal_smooth_weight * s_loss
        return {'loss': loss, 'normal_loss': n_loss, 'normal_smooth_loss': s_loss}


def fused_losses(main, normal, model_outputs, ground_truth, model_input, count):
    """MainLoss.forward + NormalLoss.forward (same branches, same dictionary keys) through ops.Stage2Losses: two launches
    forward, one backward.  Returns (total loss, MainLoss-style dict, NormalLoss-style dict) or None when the fused path
    does not apply (tensors not on the GPU, an output the kernels expect is missing): callers then use the modules."""
    from .. import ops
    o = model_outputs
    need = ('sg_rgb_values', 'normal_pred', 'normal_values', 'network_object_mask', 'object_mask')
    if any(k not in o for k in need) or not o['sg_rgb_values'].is_cuda or main.loss_type not in ('L1', 'L2'):
        return None
    ma, mb = o['network_object_mask'][0], o['object_mask'][0]
    if ma.dtype != torch.bool or mb.dtype != torch.bool:
        return None
    rgb, rgb_gt = o['sg_rgb_values'], ground_truth['rgb'].to(o['sg_rgb_values'].device)
    L, N = rgb.shape[0], rgb.shape[1]
    alb = alb_j = wgt = wgt_j = vis = vis_gt = nrm_j = None
    if 'albedo_jitter' in o and main.albedo_smooth_weight > 0:
        alb, alb_j = o['albedo_values'], o['albedo_jitter']  # [1, N, 3]: the kernels read them as [N, 3]
    if 'rough_jitter' in o and main.rough_smooth_weight > 0:
        wgt, wgt_j = o['rough_values'], o['rough_jitter']
    has_vis = 'visibility' in model_input and 'visibility' in o
    if has_vis:  # loss.py:81-87
        if 'vis_train_gt' in model_input and 'light_vis_train' in model_input and 'vis_train' in o:
            vis, vis_gt = o['vis_train'], model_input['vis_train_gt']
        elif 'light_vis_train' in model_input and 'vis_train' in o:
            vis, vis_gt = o['vis_train'], model_input['visibility']
        else:
            vis, vis_gt = o['visibility'], model_input['visibility']
        vis_gt = vis_gt.float()
    nrm, nrm_gt = o['normal_pred'], o['normal_values']
    if 'normal_jitter' in o and normal.normal_smooth_weight > 0:
        nrm_j = o['normal_jitter']
    # ``count`` may be a python int (host-known) or a device float tensor [1] (never synchronised; under data
    # parallelism the all-reduced count): the kernels then divide by it on the device
    count_dev = count.reshape(1).float() if torch.is_tensor(count) else None
    c = 1.0 if count_dev is not None else float(max(count, 1))
    nb = 1 if wgt is None else wgt.shape[-1]
    V = 1 if vis is None else vis.shape[0]
    inv = [1.0 / (c * L * 3), 1.0 / (c * 3), 1.0 / (c * nb), 1.0 / (c * V), 1.0 / (c * 3), 1.0 / (c * 3)]
    if count_dev is None and count == 0:
        inv = [0.0] * 6  # the reference returns 0 for every term of an empty mask
    w = [float(main.sg_rgb_weight), float(main.albedo_smooth_weight) if alb is not None else 0.0,
         float(main.rough_smooth_weight) if wgt is not None else 0.0, float(main.vis_weight) if vis is not None else 0.0,
         float(normal.normal_weight), float(normal.normal_smooth_weight) if nrm_j is not None else 0.0]
    total, t = ops.Stage2Losses.apply(rgb, rgb_gt, alb, alb_j, wgt, wgt_j, vis, vis_gt, nrm, nrm_gt, nrm_j, ma, mb,
                                      1 if main.loss_type == 'L2' else 0, inv, w, count_dev)
    terms = {'sg_rgb_loss': t[0], 'albedo_smooth_loss': t[1] if alb is not None else None,
             'rough_smooth_loss': t[2] if wgt is not None else None, 'loss': total}
    if has_vis:
        terms['vis_loss'] = t[3]
    terms_n = {'loss': total, 'normal_loss': t[4], 'normal_smooth_loss': t[5] if nrm_j is not None else None}
    return total, terms, terms_n
