"""Stage-2 train step: the body of TrainRunner.run (stage2/trainer.py:355-410,462-464) and the
train_fix schedule (stage2/trainer.py:485-513), without datasets / checkpoints / plots (out of scope,
SURVEY 2).  Adam on the MLPs, SparseAdam on the light direction [n,3] and intensity [n,1] embeddings
(trainer.py:126-168), per-iteration MultiStepLR."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..dist import DataParallel
from .. import ops
from ..optim import FlatAdam, RowSparseAdam
from .loss import MainLoss, NormalLoss, fused_losses


class VisPlus(object):
    """The extra visibility supervision of ``train.vis_plus`` (stage2/trainer.py:209-214 tables, :384-392 per-step
    selection): per view, the P extra directions written by stage 1 (``vis_plus/light_dir.json`` + ``vis_plus/view_XX.npy``)
    followed by the view's L_v initial (SDPS-Net) light estimates with the stage-1 visibility maps of the calibrated
    lights; every step draws ``train.vis_train_num`` of the P + L_v rows with ``np.random.choice(replace=False)``.

    ``views[v]`` = handoff.load_view(...) dictionaries ('vis_plus' [P,hw], 'vis_plus_light' [P,3], 'visibility'
    [L_v,hw]); ``light_vis_train[v]`` [L_v,3] = the FIXED initial light estimates of view v (trainer.py:149; they are
    concatenated un-normalised, :388).  All tables stay resident in HBM (P = 256: 0.3 GB per 612x512 view), so the
    per-step selection is one two-index gather on the device instead of the reference's host tensor construction +
    H2D copy of the whole [P + L_v, hw] table every iteration."""

    def __init__(self, views, light_vis_train, vnum, device, rng=None):
        self.vnum, self.device = int(vnum), device
        self.rng = rng if rng is not None else np.random
        self.lights = [torch.cat([v['vis_plus_light'].to(device).float(), lv.to(device).float()], dim=0)
                       for v, lv in zip(views, light_vis_train)]
        self.vis = [torch.cat([v['vis_plus'].to(device).float().reshape(v['vis_plus'].shape[0], -1),
                               v['visibility'].to(device).float()], dim=0) for v in views]
        for l, m in zip(self.lights, self.vis):
            assert l.shape[0] == m.shape[0]  # trainer.py:391

    def select(self, vidx, sampling_idx=None):
        """(light_vis_train [vnum,3], vis_train_gt [vnum,n]) of one step; consumes one np.random.choice draw."""
        lights, vis = self.lights[int(vidx)], self.vis[int(vidx)]
        sidx = torch.as_tensor(self.rng.choice(np.arange(lights.shape[0]), self.vnum, replace=False)).long().to(self.device)
        if sampling_idx is None:
            return lights[sidx], vis[sidx]
        return lights[sidx], vis[sidx[:, None], sampling_idx.to(self.device).long()[None, :]]


def scheduler_milestones(conf, n_views):
    """The iteration milestones of both learning-rate schedulers as stage2/trainer.py:118-121 derives them: the configuration's
    ``train.sg_sched_milestones`` count EPOCHS of the multi-light data set; one epoch is n_views items, and with ``train.multi_light``
    every item is visited ``train.light_bs`` times as often.  -> list of iterations for ``TrainStep(..., milestones=...)``."""
    ms = list(conf.get_list('train.sg_sched_milestones', default=[]))
    ms = [int(m) * int(n_views) for m in ms]
    if conf.get_bool('train.multi_light', default=False):
        ms = [m * conf.get_int('train.light_bs') for m in ms]
    return ms


class TrainStep(object):
    FUSED_LOSSES = True  # False: evaluate MainLoss / NormalLoss with the torch formulation (cross-check, other loss types)

    def __init__(self, model, conf, n_lights_total, light_init, device, milestones=(), dp=None, vis_plus=None):
        self.model, self.conf, self.device = model, conf, device
        # frozen copy of the initial light estimates = torch.cat(self.light_vis_train) of trainer.py:149,377
        self.light_vis_table = light_init.detach().clone().to(device)
        self.vis_plus = vis_plus
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
        # Adam over one flat parameter / moment / gradient allocation: one launch per step (optim.FlatAdam = torch.optim.Adam's
        # arithmetic and state layout); torch's own implementation for parameters that are not fp32 device tensors
        on_gpu = all(p.is_cuda and p.dtype == torch.float32 for p in model.parameters())
        self.sg_optimizer = (FlatAdam if on_gpu else torch.optim.Adam)(model.parameters(), lr=conf.get_float('train.sg_learning_rate'))
        self.sg_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.sg_optimizer, list(milestones), gamma=gamma)
        # dense gradients + RowSparseAdam = SparseAdam's update of the touched rows without coalesce() (optim.py)
        self.light_para = nn.Embedding(n_lights_total, 3, sparse=False).to(device)
        self.light_para.weight.data.copy_(light_init)
        # the switches of stage2/trainer.py:36-50, read with the REFERENCE's defaults (all False: a configuration that leaves a key
        # out trains what the reference would train; the seven shipped configurations set light_train / visibility / vis_loss):
        #   train.light_train = False  -> no light tables: the batch's own 'light_direction' is used as given, no 'light_vis_train'
        #                                 is injected (the visibility loss then supervises the L shading rows, loss.py:86-87), no
        #                                 SparseAdam step (trainer.py:126,368,403-408).  The tables below still exist, frozen and unread.
        #   train.ana_fixlight         -> the light tables stay frozen after the iteration-5000 switch (trainer.py:510)
        #   train.visibility without train.vis_loss -> the visibility net is frozen at iteration 0 and never released (trainer.py:498-499)
        #   train.normal_mlp and train.normal_joint -> NormalLoss is part of the step (trainer.py:42-44,397-399)
        self.light_train = conf.get_bool('train.light_train', default=False)
        self.ana_fixlight = conf.get_bool('train.ana_fixlight', default=False)
        self.visibility = conf.get_bool('train.visibility', default=False)
        self.vis_loss = self.visibility and conf.get_bool('train.vis_loss', default=False)
        self.normal_train = conf.get_bool('train.normal_mlp', default=False) and conf.get_bool('train.normal_joint', default=False)
        if vis_plus is not None and not self.light_train:
            # trainer.py:388 concatenates self.light_vis_train, which exists only under light_train (:149): the reference raises here too
            raise AttributeError("train.vis_plus needs train.light_train (the per-view initial light estimates 'light_vis_train', trainer.py:149,388)")
        # the per-light intensity table is trained only under train.light_inten_train (stage2/trainer.py:38,154-163; the
        # DiLiGenT-MV objects set it, the synthetic bunny / armadillo configurations do not: the model then shades with its scalar
        # brdf.light_intensity, renderer.py:202).  The table itself always exists (checkpoint layout); untrained it is a constant
        # that the step never reads.
        # EXPERIMENT (BASELINE configs[4] bf16 path; off unless the configuration asks): the 256 x 256 weight gradients of the visibility
        # net on the bf16 matrix pipe with split operands (hip.wgrad_precision; fp32-class results, tests/test_bf16_gpu.py)
        self.wgrad_bf16x6 = conf.get_bool('train.wgrad_bf16x6', default=False)
        # ('train.wgrad_precision' = 'bf16x6' | 'bf16x3' | 'bf16' selects the number of partial products; the flag above = 'bf16x6')
        self.wgrad_mode = conf.get_string('train.wgrad_precision', default='bf16x6' if self.wgrad_bf16x6 else 'fp32')
        self.wgrad_bf16x6 = self.wgrad_mode != 'fp32'
        self.chain_mode = conf.get_string('train.chain_precision', default='fp32')
        assert self.chain_mode in ('fp32', 'bf16x3'), self.chain_mode
        self.light_inten_train = self.light_train and conf.get_bool('train.light_inten_train', default=False)
        if not self.light_train:
            self.light_para.requires_grad_(False)
        self.light_decay = conf.get_bool('train.light_decay', default=False)  # trainer.py:40,463-464: whether the light tables' lr follows the milestones
        self.light_inten_para = nn.Embedding(n_lights_total, 1, sparse=False).to(device)
        nn.init.constant_(self.light_inten_para.weight, model.light_int)
        lr_l = conf.get_float('train.light_learning_rate', default=5e-4)
        groups = [{'params': list(self.light_para.parameters())}]
        if self.light_inten_train:
            groups.append({'params': list(self.light_inten_para.parameters()), 'lr': conf.get_float('train.light_inten_lr', default=lr_l)})
        else:
            self.light_inten_para.requires_grad_(False)
        self.light_optimizer = RowSparseAdam(groups, lr=lr_l)
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
            if self.visibility and not self.vis_loss:  # trainer.py:498-499 (there is no release at iteration 5000)
                self.model.visibility_net.eval().requires_grad_(False)
            if self.light_train:  # trainer.py:500-503
                self.light_para.requires_grad_(False)
                if self.light_inten_train:
                    self.light_inten_para.requires_grad_(False)
        elif self.cur_iter == 5000:
            (self.loss.sg_rgb_weight, self.loss.albedo_smooth_weight,
             self.loss.rough_smooth_weight, self.loss.vis_weight) = self._ori
            self.model.albedo_net.train().requires_grad_(True)
            self.model.rough_net.train().requires_grad_(True)
            if not self.ana_fixlight and self.light_train:  # trainer.py:510-513
                self.light_para.requires_grad_(True)
                if self.light_inten_train:
                    self.light_inten_para.requires_grad_(True)

    def step(self, model_input, ground_truth, l_slt, train_order=True, noise=None, vidx=None):
        """One optimisation step on (this rank's pixel slice of) a batch.  ``l_slt`` are the rows of the
        light tables used by this batch (trainer.py:367-379).

        Visibility supervision lights (trainer.py:377-392):
          * ``vidx`` given and a VisPlus table attached -> the reference's vis_plus draw for that view (the pixel
            subset is ``model_input['sampling_idx']`` when present);
          * ``model_input`` already carries 'light_vis_train' (+ 'vis_train_gt') -> a ready-made selection (synthetic
            batches, parity fixtures) is kept;
          * otherwise -> the fixed initial estimates of the batch's lights, normalised (trainer.py:377), supervised
            against ``model_input['visibility']`` (loss.py:84-85).

        The step is three device phases -- ``_fwd_bwd`` (forward, losses, backward, gradients gathered into the flat bucket),
        ``_reduce`` (the data-parallel all-reduce) and ``_optimise`` (Adam + SparseAdam launches) -- around host bookkeeping
        (train_fix, iteration count, schedulers); stage2/graph.py replays the device phases from HIP graphs."""
        if train_order:
            self.train_fix()
        self.dp.new_step()
        model_input = self.select_vis_lights(model_input, vidx)
        terms, out, trainable, train_light = self._fwd_bwd(model_input, ground_truth, l_slt, noise=noise)
        self._reduce(trainable)
        self._optimise(l_slt, trainable, train_light)
        self._advance(train_light)
        return terms, out

    def select_vis_lights(self, model_input, vidx):
        """The vis_plus draw of a step (host RNG + two gathers from the resident tables); other batches pass through."""
        if self.vis_plus is None or vidx is None:
            return model_input
        model_input = dict(model_input)
        # under data parallelism 'sampling_idx' is this rank's pixel slice (dist.shard_stage2) and every rank draws
        # the same rows from an identically seeded np.random stream
        sidx = model_input.get('sampling_idx')
        if sidx is not None and sidx.dim() == 2:
            sidx = sidx[0]  # batch dimension of the collated sample (trainer.py:392)
        model_input['light_vis_train'], model_input['vis_train_gt'] = self.vis_plus.select(vidx, sidx)
        return model_input

    def _fwd_bwd(self, model_input, ground_truth, l_slt, noise=None, count=None):
        if self.chain_mode != 'fp32':
            # EXPERIMENT (conf train.chain_precision = 'bf16x3'): the backward chains of the 256-wide networks (the V supervised rows of
            # visibility_net, the normal / albedo networks) as three bf16 partial products (ops.chain_precision)
            with ops.chain_precision(self.chain_mode):
                return self._fwd_bwd_exact(model_input, ground_truth, l_slt, noise, count)
        return self._fwd_bwd_exact(model_input, ground_truth, l_slt, noise, count)

    def _fwd_bwd_exact(self, model_input, ground_truth, l_slt, noise=None, count=None):
        """Forward, losses, backward; under data parallelism the gradients end up in the flat bucket (not yet reduced).
        ``count``: the (global) masked-pixel count as a device tensor [1] when the caller has already formed it (graph replay
        under data parallelism: the count's all-reduce stays outside the captured graph)."""
        model_input = dict(model_input)
        if not self.light_train:
            pass  # ground-truth lights: 'light_direction' (and the model's scalar intensity) as the batch brings them (trainer.py:368)
        elif l_slt.is_cuda and self.light_para.weight.is_cuda:
            # both table lookups + the normalisation in one launch (one more for the dense table gradients in backward)
            d_, i_ = ops.LightRows.apply(self.light_para.weight, self.light_inten_para.weight, l_slt.long())
            model_input['light_direction'] = d_
            if self.light_inten_train:  # (trainer.py:378-379; otherwise the model's scalar light_int)
                model_input['light_intensity'] = i_
        else:
            model_input['light_direction'] = F.normalize(self.light_para(l_slt), p=2, dim=-1)
            if self.light_inten_train:
                model_input['light_intensity'] = self.light_inten_para(l_slt)
        if self.light_train and 'light_vis_train' not in model_input:
            if l_slt.is_cuda and self.light_vis_table.is_cuda:
                from .. import hip
                model_input['light_vis_train'] = hip.light_rows_fwd(self.light_vis_table.contiguous(), None, l_slt.long().contiguous())[0]
            else:
                model_input['light_vis_train'] = F.normalize(self.light_vis_table[l_slt], p=2, dim=-1)
        # ONE mask count per step (a tiny all-reduce under data parallelism), shared by every loss term.  Both masks
        # are inputs (the model passes them through as 'network_object_mask' / 'object_mask'), so the count -- a host
        # synchronisation -- is taken BEFORE the forward pass: afterwards the host runs ahead of the GPU through
        # forward, losses and backward instead of stalling behind the 22 ms visibility launch.
        sm, om = model_input['surface_mask'], model_input['object_mask']
        if count is not None:
            pass
        elif self.FUSED_LOSSES and sm.is_cuda and self.normal_train:
            # the count stays on the device (all-reduced there under data parallelism): the fused loss kernels divide by
            # it, so the step has no host synchronisation of its own (the model's only one is the surface-pixel list,
            # and a batch may bring that along as 'surface_idx')
            count = self.dp.masked_count_tensor(sm, om)
        else:
            # (the MainLoss-only configuration and CPU tensors: the torch loss formulation divides by a host value)
            if sm.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('TrainStep: this configuration (no jointly trained normal net: train.normal_mlp / train.normal_joint off) takes the '
                                   'torch loss formulation, whose mask count is a host value -- the step cannot be captured into a HIP graph; run it eagerly')
            count = self.dp.global_count(sm & om)
        out = self.model(model_input, noise=noise)
        fl = fused_losses(self.loss, self.loss_n, out, ground_truth, model_input, count) if (self.FUSED_LOSSES and self.normal_train) else None
        if fl is None and self.FUSED_LOSSES and self.normal_train:
            ops.fallback('stage2 losses -> torch formulation of MainLoss / NormalLoss', out['sg_rgb_values'], 'loss type or outputs outside csrc/loss.hip')
        if fl is None and torch.is_tensor(count):
            count = int(count.item())
        if fl is not None:  # both loss modules in two launches forward / one backward (csrc/loss.hip)
            loss, terms, terms_n = fl
            terms = dict(terms)
        elif not self.normal_train:
            # trainer.py:397-399: without a jointly trained normal net the step has MainLoss only (no shipped configuration; torch formulation)
            terms = dict(self.loss(out, ground_truth, model_input, count=count))
            terms_n = {'loss': None, 'normal_loss': None, 'normal_smooth_loss': None}
            loss = terms['loss']
        else:
            terms = dict(self.loss(out, ground_truth, model_input, count=count))
            terms_n = self.loss_n(out, count=count)
            loss = terms['loss'] + terms_n['loss']
        train_light = self.light_para.weight.requires_grad
        trainable = None
        if self.dp.enabled:
            trainable = [p for p in self.model.parameters() if p.requires_grad] \
                + (([self.light_para.weight] + ([self.light_inten_para.weight] if self.light_inten_train else [])) if train_light else [])
            # every .grad dropped; gather_grads collects what autograd hands over into the flat bucket (one multi-tensor copy)
            self.dp.prepare_grads(list(self.model.parameters()) + [self.light_para.weight, self.light_inten_para.weight])
        else:
            if isinstance(self.sg_optimizer, FlatAdam):
                self.sg_optimizer.attach_grads()  # grads dropped (no launch); step() gathers what autograd hands over with one multi-tensor copy
            else:
                self.sg_optimizer.zero_grad()
            if train_light:
                self.light_optimizer.zero_grad()
        if loss.requires_grad:
            # (the seed of the backward pass is a cached constant: backward() alone launches a ones_like fill per step)
            seed = getattr(self, '_grad_seed', None)
            if seed is None or seed.device != loss.device or seed.shape != loss.shape:
                seed = self._grad_seed = torch.ones_like(loss)
            if self.wgrad_bf16x6:
                from .. import hip
                with hip.wgrad_precision(self.wgrad_mode):
                    loss.backward(seed)
            else:
                loss.backward(seed)
        # (a rank whose pixel slice has no surface pixel gets constants from the model: its loss has no graph, it skips
        # backward and contributes the zero-filled bucket, so the other ranks never wait for a collective it left out)
        if self.dp.enabled:
            self.dp.gather_grads(trainable)
        terms['total'] = loss
        terms['normal_loss'] = terms_n['normal_loss']
        return terms, out, trainable, train_light

    def _reduce(self, trainable):
        if self.dp.enabled:
            self.dp.allreduce_bucket(trainable)

    def _optimise(self, l_slt, trainable, train_light):
        self.sg_optimizer.step()
        if hasattr(self.model, 'invalidate_packs'):
            # packs are keyed on the parameters' version counters; fused / capturable optimiser implementations do not bump them
            self.model.invalidate_packs(trainable_only=True)
        if train_light:
            self.light_optimizer.step(rows=l_slt)

    def _advance(self, train_light):
        self.cur_iter += 1
        self.sg_scheduler.step()
        if train_light and self.light_decay:
            self.light_scheduler.step()


def psnr(img1, img2, mask=None):
    """stage2/trainer.py:268-276."""
    if mask is not None:
        m = mask.to(torch.bool)[:, None]
        img1, img2 = torch.masked_select(img1, m).view(-1, 3), torch.masked_select(img2, m).view(-1, 3)
    mse = ((img1 - img2) ** 2).mean()
    return 100.0 if mse == 0 else float(-10.0 * torch.log10(mse))


def mae_error(img1, img2, mask=None, normalize=True):
    """stage2/trainer.py:257-266: mean angular error in degrees between two normal sets [N,3] (masked), and the per-pixel errors."""
    if mask is not None:
        m = mask.to(torch.bool)[:, None]
        img1, img2 = torch.masked_select(img1, m).view(-1, 3), torch.masked_select(img2, m).view(-1, 3)
    if normalize:
        img1, img2 = F.normalize(img1, dim=-1), F.normalize(img2, dim=-1)
    err = torch.acos((img1 * img2).sum(-1).clamp(-1, 1)) * 180.0 / np.pi
    return err.mean(), err


@torch.no_grad()
def validate_view(model, model_input, rgb_gt, pixel_chunk=None):
    """The validation half of TrainRunner.plot_to_disk (stage2/trainer.py:278-326; the plotting half, utils/plots.py, is file output):
    one view under ONE light rendered whole (``pixel_chunk`` None; the reference: 1024-pixel chunks through split_input /
    merge_output) -> {'psnr': PSNR of sg_rgb_values over network_object_mask & object_mask (:312-313), 'normal_MAE' and the
    'normal_mae' error map when the model has a normal net and the input carries 'gt_normal' (:315-320), 'outputs': the merged
    model outputs}.  ``rgb_gt`` [N, 3] (the picked light's image), ``model_input['light_direction']`` [1, 3]."""
    from .relight import merge_output, split_input
    n = model_input['uv'].shape[1]
    chunks = [model_input] if pixel_chunk is None else split_input(model_input, n, pixel_chunk)
    res = [{k: v.detach() for k, v in model(s).items() if torch.is_tensor(v)} for s in chunks]
    out = merge_output(res, n, 1) if len(res) > 1 else {k: (v.reshape(-1) if v.dim() < 3 else v.reshape(-1, v.shape[-1])) for k, v in res[0].items()}
    mask = torch.logical_and(out['network_object_mask'], out['object_mask'])
    rep = {'psnr': psnr(rgb_gt.to(out['sg_rgb_values'].device), out['sg_rgb_values'], mask), 'outputs': out}
    if 'normal_pred' in out and 'gt_normal' in model_input:
        mae, err = mae_error(model_input['gt_normal'][0], out['normal_pred'], mask)
        emap = torch.zeros_like(out['normal_pred'][..., 0])
        emap[mask] = err
        rep['normal_MAE'], rep['normal_mae'] = float(mae), emap
    return rep
