"""Stage-2 losses with the reference's interface (stage2/model/loss.py:6-141), device-agnostic
(the reference hard-codes .cuda()).  Tiny reductions over masked pixels -- host-side torch ops.

Under ray/pixel data parallelism the masked MEANS must use GLOBAL element counts so that the summed
rank gradients equal the single-GPU gradient; ``count_scale`` carries
(local count / global count * world_size) for that purpose (1.0 on a single GPU)."""
import torch
from torch import nn
from torch.nn import functional as F


def _mean(fn_sum, a, b, denom):
    return fn_sum(a, b) / denom


class MainLoss(nn.Module):
    def __init__(self, sg_rgb_weight, loss_type='L1', albedo_smooth_weight=0, rough_smooth_weight=0, vis_weight=1.0):
        super().__init__()
        self.sg_rgb_weight = sg_rgb_weight
        self.albedo_smooth_weight = albedo_smooth_weight
        self.rough_smooth_weight = rough_smooth_weight
        self.vis_weight = vis_weight
        if loss_type not in ('L1', 'L2'):
            raise Exception('Unknown loss_type!')
        self.loss_type = loss_type
        self.global_count = None  # set by psnerf_amd.dist for data-parallel runs

    def _img(self, a, b, n_global=None):
        d = (a - b).abs() if self.loss_type == 'L1' else (a - b) ** 2
        return d.sum() / (d.numel() if n_global is None else n_global)

    def _l1(self, a, b, n_global=None):
        d = (a - b).abs()
        return d.sum() / (d.numel() if n_global is None else n_global)

    def _n(self, mask, per_pixel):
        """global number of elements behind a masked mean (None on a single GPU)."""
        if self.global_count is None:
            return None
        return self.global_count(mask) * per_pixel

    def forward(self, model_outputs, ground_truth, model_input=None):
        m, om = model_outputs['network_object_mask'], model_outputs['object_mask']
        mask = m & om
        dev = mask.device
        empty = int(mask.sum()) == 0 if self.global_count is None else self.global_count(mask) == 0
        zero = torch.tensor(0.0, device=dev)
        rgb_gt = ground_truth['rgb'].to(dev)
        if empty:
            rgb_loss = zero
        else:
            mk = mask.expand(rgb_gt.shape[0], -1)
            rgb_loss = self._img(model_outputs['sg_rgb_values'][mk].reshape(-1, 3), rgb_gt[mk].reshape(-1, 3),
                                 self._n(mask, 3 * rgb_gt.shape[0]))
        loss = self.sg_rgb_weight * rgb_loss
        a_loss = r_loss = None
        if 'albedo_jitter' in model_outputs and self.albedo_smooth_weight > 0:
            x, xj = model_outputs['albedo_values'], model_outputs['albedo_jitter']
            mk = mask.expand(x.shape[0], -1)
            a_loss = zero if empty else self._l1(x[mk], xj[mk], self._n(mask, x.shape[-1] * x.shape[0]))
            loss = loss + self.albedo_smooth_weight * a_loss
        if 'rough_jitter' in model_outputs and self.rough_smooth_weight > 0:
            x, xj = model_outputs['rough_values'], model_outputs['rough_jitter']
            mk = mask.expand(x.shape[0], -1)
            r_loss = zero if empty else self._l1(x[mk], xj[mk], self._n(mask, x.shape[-1] * x.shape[0]))
            loss = loss + self.rough_smooth_weight * r_loss
        terms = {'sg_rgb_loss': rgb_loss, 'albedo_smooth_loss': a_loss, 'rough_smooth_loss': r_loss}
        if 'visibility' in model_input and 'visibility' in model_outputs:
            if 'vis_train_gt' in model_input and 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['vis_train_gt']
            elif 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['visibility']
            else:
                v, gt = model_outputs['visibility'][..., 0], model_input['visibility']
            if empty:
                vis_loss = zero
            else:
                mk = mask.expand(gt.shape[0], -1)
                vis_loss = self._img(v[mk].reshape(-1), gt[mk].reshape(-1), self._n(mask, gt.shape[0]))
            loss = loss + self.vis_weight * vis_loss
            terms['vis_loss'] = vis_loss
        terms['loss'] = loss
        return terms


class NormalLoss(nn.Module):
    def __init__(self, normal_weight, normal_smooth_weight=0):
        super().__init__()
        self.normal_weight, self.normal_smooth_weight = normal_weight, normal_smooth_weight
        self.global_count = None

    def forward(self, model_outputs):
        gt = F.normalize(model_outputs['normal_values'], dim=-1)
        mask = model_outputs['network_object_mask'] & model_outputs['object_mask']
        dev = mask.device
        n_glob = None if self.global_count is None else self.global_count(mask)
        empty = int(mask.sum()) == 0 if n_glob is None else n_glob == 0
        zero = torch.tensor(0.0, device=dev)
        if empty:
            n_loss = zero
        else:
            d = (model_outputs['normal_pred'][mask].reshape(-1, 3) - gt[mask].reshape(-1, 3)) ** 2
            n_loss = d.sum() / (d.numel() if n_glob is None else n_glob * 3)
        loss = self.normal_weight * n_loss
        s_loss = None
        if 'normal_jitter' in model_outputs and self.normal_smooth_weight > 0:
            if empty:
                s_loss = zero
            else:
                d = (model_outputs['normal_pred'][mask] - model_outputs['normal_jitter'][mask]).abs()
                s_loss = d.sum() / (d.numel() if n_glob is None else n_glob * 3)
            loss = loss + self.normal_smooth_weight * s_loss
        return {'loss': loss, 'normal_loss': n_loss, 'normal_smooth_loss': s_loss}
