"""Stage-2 train step: the body of TrainRunner.run (stage2/trainer.py:355-410,462-464) and the
train_fix schedule (stage2/trainer.py:485-513), without datasets / checkpoints / plots (out of scope,
SURVEY 2).  Adam on the MLPs, SparseAdam on the light direction [n,3] and intensity [n,1] embeddings
(trainer.py:126-168), per-iteration MultiStepLR."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..dist import DataParallel
from ..optim import RowSparseAdam
from .loss import MainLoss, NormalLoss


class TrainStep(object):
    def __init__(self, model, conf, n_lights_total, light_init, device, milestones=(), dp=None):
        self.model, self.conf, self.device = model, conf, device
        lk = dict(sg_rgb_weight=1.0, loss_type='L1', albedo_smooth_weight=0.05, rough_smooth_weight=0.01, vis_weight=1)
        lc = conf.get_config('loss', default=None) if hasattr(conf, 'get_config') else None
        if lc:
            lk.update({k: lc[k] for k in lk if k in lc})
        self.loss = MainLoss(**lk)
        nk = dict(normal_weight=1, normal_smooth_weight=0.05)
        nc = conf.get_config('normal.loss', default=None) if hasattr(conf, 'get_config') else None
        if nc:
            nk.update({k: nc[k] for k in nk if k in nc})
        self.loss_n = NormalLoss(**nk)
        self.dp = dp if dp is not None else DataParallel(device)
        if self.dp.enabled:
            self.loss.global_count = self.dp.global_count
            self.loss_n.global_count = self.dp.global_count
        gamma = conf.get_float('train.sg_sched_factor', default=0.0)
        self.sg_optimizer = torch.optim.Adam(model.parameters(), lr=conf.get_float('train.sg_learning_rate'))
        self.sg_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.sg_optimizer, list(milestones), gamma=gamma)
        # dense gradients + RowSparseAdam = SparseAdam's update of the touched rows without coalesce() (optim.py)
        self.light_para = nn.Embedding(n_lights_total, 3, sparse=False).to(device)
        self.light_para.weight.data.copy_(light_init)
        self.light_inten_para = nn.Embedding(n_lights_total, 1, sparse=False).to(device)
        nn.init.constant_(self.light_inten_para.weight, model.light_int)
        lr_l = conf.get_float('train.light_learning_rate', default=5e-4)
        self.light_optimizer = RowSparseAdam(
            [{'params': list(self.light_para.parameters())},
             {'params': list(self.light_inten_para.parameters()),
              'lr': conf.get_float('train.light_inten_lr', default=lr_l)}], lr=lr_l)
        self.light_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.light_optimizer, list(milestones), gamma=gamma)
        self.cur_iter = 0

    def train_fix(self):  # trainer.py:485-513
        if self.cur_iter == 0:
            self._ori = (self.loss.sg_rgb_weight, self.loss.albedo_smooth_weight,
                         self.loss.rough_smooth_weight, self.loss.vis_weight)
            self.loss.sg_rgb_weight = 0
            self.loss.albedo_smooth_weight = 0
            self.loss.rough_smooth_weight = 0
            self.loss.vis_weight = 10
            self.model.albedo_net.eval().requires_grad_(False)
            self.model.rough_net.eval().requires_grad_(False)
            self.light_para.requires_grad_(False)
            self.light_inten_para.requires_grad_(False)
        elif self.cur_iter == 5000:
            (self.loss.sg_rgb_weight, self.loss.albedo_smooth_weight,
             self.loss.rough_smooth_weight, self.loss.vis_weight) = self._ori
            self.model.albedo_net.train().requires_grad_(True)
            self.model.rough_net.train().requires_grad_(True)
            self.light_para.requires_grad_(True)
            self.light_inten_para.requires_grad_(True)

    def step(self, model_input, ground_truth, l_slt, train_order=True, noise=None):
        """One optimisation step on (this rank's pixel slice of) a batch.  ``l_slt`` are the rows of the
        light tables used by this batch (trainer.py:367-379)."""
        if train_order:
            self.train_fix()
        self.dp.new_step()
        model_input = dict(model_input)
        model_input['light_direction'] = F.normalize(self.light_para(l_slt), p=2, dim=-1)
        model_input['light_intensity'] = self.light_inten_para(l_slt)
        # ONE mask count per step (a tiny all-reduce under data parallelism), shared by every loss term.  Both masks
        # are inputs (the model passes them through as 'network_object_mask' / 'object_mask'), so the count -- a host
        # synchronisation -- is taken BEFORE the forward pass: afterwards the host runs ahead of the GPU through
        # forward, losses and backward instead of stalling behind the 22 ms visibility launch.
        count = self.dp.global_count(model_input['surface_mask'] & model_input['object_mask'])
        out = self.model(model_input, noise=noise)
        terms = dict(self.loss(out, ground_truth, model_input, count=count))
        terms_n = self.loss_n(out, count=count)
        loss = terms['loss'] + terms_n['loss']
        self.sg_optimizer.zero_grad()
        train_light = self.light_para.weight.requires_grad
        if train_light:
            self.light_optimizer.zero_grad()
        loss.backward()
        self.dp.allreduce_grads([p for p in self.model.parameters() if p.requires_grad]
                                + ([self.light_para.weight, self.light_inten_para.weight] if train_light else []))
        self.sg_optimizer.step()
        if train_light:
            self.light_optimizer.step(rows=l_slt)
        self.cur_iter += 1
        self.sg_scheduler.step()
        if train_light:
            self.light_scheduler.step()
        terms['total'] = loss
        terms['normal_loss'] = terms_n['normal_loss']
        return terms, out


def psnr(img1, img2, mask=None):
    """stage2/trainer.py:268-276."""
    if mask is not None:
        m = mask.to(torch.bool)[:, None]
        img1, img2 = torch.masked_select(img1, m).view(-1, 3), torch.masked_select(img2, m).view(-1, 3)
    mse = ((img1 - img2) ** 2).mean()
    return 100.0 if mse == 0 else float(-10.0 * torch.log10(mse))
