"""Stage-1 train step with the reference's ``Trainer`` interface (stage1/model/training.py:20-60,
120-198; visualisation is out of scope).  Under data parallelism every rank renders a slice of the
sampled pixels of the same view and the gradients are summed with one flat-bucket all-reduce."""
import contextlib

import numpy as np
import torch

from .. import hip, ops
from ..dist import DataParallel
from ..optim import FlatAdam
from .losses import Loss
from .rendering import is_per_ray_noise


def gather_pixels(img, pix):
    """stage1/model/common.py:172-202: nearest grid_sample with the reference's 2p/w - 1 scaling."""
    _, _, h, w = img.shape
    p = pix.clone().detach()
    p[:, :, 0] = 2.0 * p[:, :, 0] / w - 1
    p[:, :, 1] = 2.0 * p[:, :, 1] / h - 1
    v = torch.nn.functional.grid_sample(img, p.unsqueeze(1), mode='nearest', align_corners=True)
    return v.squeeze(2).detach().permute(0, 2, 1)


class Trainer(object):
    def __init__(self, model, optimizer, cfg_all, device=None, dp=None, **kwargs):
        cfg = cfg_all['training']
        self.model, self.optimizer, self.device, self.cfg = model, optimizer, device, cfg
        self.n_training_points = cfg['n_training_points']
        self.n_eval_points = cfg['n_training_points']  # training.py:27 reads the same key
        self.normal_loss = cfg.get('normal_loss', False)
        self.normal_after = cfg.get('normal_after', -1)
        self.angle = cfg.get('normal_angle', None)
        self.mask_loss = cfg.get('mask_loss', False)
        self.rendering_technique = cfg['type']
        self.loss = Loss(cfg['lambda_l1_rgb'], cfg['lambda_normals'], cfg.get('lambda_normloss', 1.0),
                         cfg.get('lambda_mask', 1.0), device=device)
        self.dp = dp if dp is not None else DataParallel(device)
        if self.dp.enabled:
            self.loss.global_sum = self.dp.global_sum_int
            self.loss.global_count = self.dp.global_count_tensor
            self.loss.global_rays = lambda: self.dp.global_rays
            self.loss.global_counts = self.dp.all_reduce_sum_
        # training forward without host synchronisation (Renderer._unisurf_sync_free); a caller that injects noise or
        # needs the reference-shaped out_dict (compact diff_norm) gets the reference-shaped path
        self.sync_free = bool(kwargs.get('sync_free', True))
        # EXPERIMENT (BASELINE configs[4] bf16 path): the 256 x 256 weight gradients on the bf16 matrix pipe with split operands
        self.wgrad_bf16x6 = bool(cfg.get('wgrad_bf16x6', False))
        self.wgrad_mode = cfg.get('wgrad_precision', 'bf16x6' if self.wgrad_bf16x6 else None)   # 'bf16x6' | 'bf16x3' | 'bf16' | None = the process-wide setting
        # ... and the matrix work of the geometry / appearance chains as three bf16 partial products (ops.chain_precision): 'bf16x3' | None
        self.chain_mode = cfg.get('chain_precision', None)

    def train_step(self, data, it=None, pix=None, noise=None):
        self.model.train()
        if self.dp.enabled:
            trainable = [p for p in self.model.parameters() if p.requires_grad]
            # every .grad dropped; allreduce_grads gathers what autograd hands over into the flat bucket (one multi-tensor copy)
            self.dp.prepare_grads(list(self.model.parameters()))
        elif isinstance(self.optimizer, FlatAdam):
            self.optimizer.attach_grads()  # grads dropped (no launch); step() gathers what autograd hands over with one multi-tensor copy
        else:
            self.optimizer.zero_grad()
        with contextlib.ExitStack() as modes:
            if self.chain_mode is not None:
                modes.enter_context(ops.chain_precision(self.chain_mode))
            terms = self.compute_loss(data, it=it, pix=pix, noise=noise)
            if terms['loss'].requires_grad:
                if self.wgrad_mode is not None:
                    modes.enter_context(hip.wgrad_precision(self.wgrad_mode))
                terms['loss'].backward()
        if self.dp.enabled:
            self.dp.allreduce_grads(trainable)
        self.optimizer.step()
        net = getattr(self.model, 'model', None)
        if hasattr(net, 'invalidate_packs'):
            # the weight packs are keyed on the parameters' version counters, which a fused / capturable optimiser
            # implementation (torch.optim.Adam(fused=True): verified) does NOT bump: after a step they are stale by definition
            net.invalidate_packs()
        if hasattr(net, 'prepack'):
            net.prepack(occupancy=True)  # the next step's ray march starts with this pack: queue it behind the backward
        return terms

    @torch.no_grad()
    def render_visdata(self, data_loader, it, out_render_path, chunk=65536):
        """The periodic image grid of stage1/train.py (stage1/model/training.py:62-118): for the first two items of ``data_loader``
        the panels [input image | rendered rgb | predicted normal (camera frame, /2 + 0.5) | (with a normal map: its normals | the
        angular error / 45 deg through the 'jet' colour map on mask | mask_pred) | mask_pred | acc | phong preview] side by side, items
        stacked vertically, written as an RGB image to ``out_render_path`` (PIL) and returned as a uint8 array [H, W, 3].
        Whole images are rendered through 'unisurf' (eval_) and 'phong_renderer' on the int64 x-major pixel grid in chunks of
        ``chunk`` pixels (the reference: 1024; per-pixel results do not depend on the chunking)."""
        from ..handoff import arange_pixels, to_hw
        from ..metrics import MAE
        to_img = lambda x: (np.asarray(x).astype(np.float32).clip(0, 1) * 255).astype(np.uint8)
        to_np = lambda x: x.detach().cpu().numpy()
        self.model.eval()
        rows = []
        for di, data in enumerate(data_loader):
            if di >= 2:
                break
            (img, mask, world_mat, camera_mat, scale_mat, img_idx, normal, norm_mask, mask_valid) = self.process_data_dict(data)
            h, w = img.shape[-2:]
            ploc = arange_pixels(h, w, self.device)
            panels = [to_img(to_np(img))[0].transpose(1, 2, 0)]
            outs = [self.model(px, camera_mat, world_mat, scale_mat, 'unisurf', add_noise=False, eval_=True, it=it)
                    for px in torch.split(ploc, int(chunk), dim=1)]
            rgb_pred = to_np(to_hw(torch.cat([o['rgb'] for o in outs], dim=1), h, w))
            panels.append(to_img(rgb_pred))
            mask_pred = to_np(to_hw(torch.cat([o['mask_pred'] for o in outs], dim=0), h, w)).repeat(3, axis=-1)
            norm_pred = to_np(to_hw(torch.cat([o['normal_pred'] for o in outs], dim=1), h, w))
            norm_pred = np.einsum('ij,hwi->hwj', to_np(world_mat)[0, :3, :3] * np.array([[1, -1, -1]]), norm_pred)   # training.py:92
            panels.append(to_img(norm_pred / 2. + 0.5))
            if normal is not None:
                n_gt = to_np(normal[0].permute(1, 2, 0))
                panels.append(to_img(n_gt / 2. + 0.5))
                error = MAE(norm_pred, n_gt.clip(-1, 1))[1] / 45
                try:
                    import matplotlib.pyplot as plt
                    cm = plt.get_cmap('jet')
                except Exception:  # noqa: BLE001  (matplotlib is the reference's dependency for this one panel: grey levels without it)
                    cm = lambda e: np.repeat(np.asarray(e)[..., None], 4, axis=-1)
                panels.append(to_img(cm(error.clip(0, 1) * (to_np(mask.bool())[0, 0] | mask_pred[..., 0]))[..., :3]))
            panels.append(to_img(mask_pred))
            mask_acc = to_np(to_hw(torch.cat([o['acc_map'] for o in outs], dim=1), h, w)).repeat(3, axis=-1)
            panels.append(to_img(mask_acc))
            phong = torch.cat([self.model(px, camera_mat, world_mat, scale_mat, 'phong_renderer', add_noise=False, eval_=True, it=it)['rgb']
                               for px in torch.split(ploc, int(chunk), dim=1)], dim=1)
            panels.append(to_img(to_np(to_hw(phong, h, w))))
            rows.append(np.concatenate(panels, axis=-2))
        grid = np.concatenate(rows, axis=0).astype(np.uint8)
        if out_render_path is not None:
            from PIL import Image
            Image.fromarray(grid).convert('RGB').save(out_render_path)
        self.model.train()
        return grid

    def process_data_dict(self, data):
        """stage1/model/training.py:120-139: the tensors of a data-loader item on the trainer's device, in the reference's
        order (img, mask_img, world_mat, camera_mat, scale_mat, img_idx, normal, norm_mask, mask_valid); absent masks are
        all ones, normal / norm_mask are None without the normal loss."""
        device = self.device
        img = data.get('img').to(device)
        img_idx = data.get('img.idx')
        batch_size, _, h, w = img.shape
        mask_img = data.get('img.mask', torch.ones(batch_size, h, w)).unsqueeze(1).to(device)
        world_mat = data.get('img.world_mat').to(device)
        camera_mat = data.get('img.camera_mat').to(device)
        scale_mat = data.get('img.scale_mat').to(device)
        normal = data.get('img.normal').to(device) if self.normal_loss else None
        norm_mask = data.get('img.norm_mask').unsqueeze(1).to(device) if self.normal_loss else None
        mask_valid = data.get('img.mask_valid', torch.ones(batch_size, h, w)).unsqueeze(1).to(device)
        return (img, mask_img, world_mat, camera_mat, scale_mat, img_idx, normal, norm_mask, mask_valid)

    def _upload(self, t):
        """Host tensor -> device without stalling the host behind the stream: a pageable source makes the copy wait for
        everything queued before it (the whole previous step), after which the host trails the GPU by its launch latency for
        the first ~0.3 ms of every step.  Two pinned staging buffers, used alternately; a buffer is rewritten only after the
        copy that last read it has completed (two steps back: never waits in practice)."""
        dev = torch.device(self.device)
        if t.is_cuda or dev.type != 'cuda':
            return t.to(dev)
        slot = self._pin_turn = 1 - getattr(self, '_pin_turn', 0)
        bufs = self.__dict__.setdefault('_pin_bufs', [None, None])
        buf, ev = bufs[slot] if bufs[slot] is not None else (None, None)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf, ev = torch.empty(t.shape, dtype=t.dtype).pin_memory(), torch.cuda.Event()
        else:
            ev.synchronize()
        buf.copy_(t)
        out = buf.to(dev, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        bufs[slot] = (buf, ev)
        return out

    FUSED_TARGETS = True  # False: the torch formulation of the ground-truth lookups (tests compare the two)

    def _targets_torch(self, img, mask_img, mask_valid, normal, norm_mask, world_mat, pix, want_normal):
        """training.py:166-191 as torch ops (any device, any batch size)."""
        B, dev = img.shape[0], img.device
        mask_gt = gather_pixels(mask_img, pix).bool().reshape(B, -1).to(torch.float32)
        mask_valid = gather_pixels(mask_valid * 1.0, pix).bool().reshape(B, -1)
        norm_mask_gt = gather_pixels(norm_mask, pix).bool().squeeze(-1) if self.normal_loss else None
        rgb_gt = gather_pixels(img, pix)
        normal_gt = None
        if want_normal:
            normal_gt = gather_pixels(normal, pix)
            if self.angle is not None:
                norm_mask_gt[normal_gt[..., -1] < np.cos(np.deg2rad(self.angle))] = False
            flip = torch.ones(1, 1, 3, device=dev)
            flip[..., 1:] = -1.0  # (1, -1, -1) without a host-to-device copy
            Rf = world_mat[:, :3, :3] * flip  # rotation as broadcast products (no library GEMM on the path)
            normal_gt = (normal_gt[..., 0:1] * Rf[:, None, :, 0] + normal_gt[..., 1:2] * Rf[:, None, :, 1]
                         + normal_gt[..., 2:3] * Rf[:, None, :, 2])
        return rgb_gt, mask_gt, mask_valid, norm_mask_gt, normal_gt

    def compute_loss(self, data, eval_mode=False, it=None, pix=None, noise=None):
        """training.py:141-198.  ``eval_mode`` renders with ``eval_=True`` (no neighbour points, ``diff_norm`` None, hence
        no smoothness term) on ``n_eval_points`` pixels.  ``pix`` / ``noise`` inject the random draws (tests)."""
        dev = self.device
        img = data['img'].to(dev)
        B, _, h, w = img.shape
        n_points = self.n_eval_points if eval_mode else self.n_training_points
        m_in = data.get('img.mask')
        assert (m_in is None or (h, w) == tuple(m_in.shape[-2:])) and n_points > 0  # training.py:156
        if pix is None and n_points >= h * w:
            # training.py:159-165 (whole image: int64 x-major arange_pixels, masks reshaped row-major).  The reference cannot
            # complete this branch: after the forward pass get_tensor_values (common.py:195) hands the int64 pixel grid to
            # grid_sample, which raises.  Same exception type and message here, before the forward instead of after it.
            raise RuntimeError('expected scalar type Float but found Long (n_training_points >= h*w selects the full-image '
                               'branch of compute_loss, stage1/model/training.py:159-165, whose int64 arange_pixels grid the '
                               'reference passes to grid_sample, common.py:195; render whole images through '
                               'handoff.arange_pixels chunks as training.py:74-90 / shape_extract.py:112-139 do)')
        def on_dev(key):  # a missing mask is all ones, built ON the device (a host default would be 1.25 MB uploaded per step)
            t = data.get(key)
            return torch.ones(B, h, w, device=dev) if t is None else t.to(dev)
        world_mat, camera_mat, scale_mat = (data['img.world_mat'].to(dev), data['img.camera_mat'].to(dev),
                                            data['img.scale_mat'].to(dev))
        normal = data.get('img.normal').to(dev) if self.normal_loss else None
        norm_mask = data.get('img.norm_mask').unsqueeze(1).to(dev) if self.normal_loss else None
        if pix is None:  # stage1/model/common.py:32-36: x then y, CPU randint
            n = int(n_points)
            px = torch.randint(0, w, size=(B, n, 1)).float()
            py = torch.randint(0, h, size=(B, n, 1)).float()
            pix = torch.cat([px, py], dim=-1)
        pix = self._upload(pix)
        if self.dp.enabled:
            noise = self.dp.shard_ray_noise(noise, pix.shape[1])
            pix = self.dp.shard_rays(pix)
        if self.rendering_technique == 'unisurf' and hasattr(self.model, 'prefetch_surface'):
            self.model.prefetch_surface(pix, camera_mat, world_mat)  # the ray-march sweep runs under the host work below
        want_normal = bool(self.normal_loss and it >= self.normal_after)
        if self.FUSED_TARGETS and pix.is_cuda and B == 1 and pix.dtype == torch.float32 and pix.is_contiguous():
            # every ground-truth lookup of the batch in one launch (psn_stage1_targets) instead of ~55
            def plane(key):
                t = data.get(key)
                return None if t is None else t.to(dev).reshape(h, w).float().contiguous()
            cos_t = float(np.cos(np.deg2rad(self.angle))) if (want_normal and self.angle is not None) else None
            rgb_gt, mask_gt, mask_valid, norm_mask_gt, normal_gt = hip.stage1_targets(
                pix[0], img[0].float().contiguous(), plane('img.mask'), plane('img.mask_valid'),
                normal[0].float().contiguous() if want_normal else None, norm_mask.reshape(h, w).float().contiguous() if self.normal_loss else None,
                world_mat[0].float().contiguous() if want_normal else None, cos_t, want_normal)
            rgb_gt, mask_gt, mask_valid = rgb_gt.unsqueeze(0), mask_gt.unsqueeze(0), mask_valid.unsqueeze(0)
            norm_mask_gt = norm_mask_gt.unsqueeze(0) if norm_mask_gt is not None else None
            normal_gt = normal_gt.unsqueeze(0) if normal_gt is not None else None
        else:
            rgb_gt, mask_gt, mask_valid, norm_mask_gt, normal_gt = self._targets_torch(
                img, on_dev('img.mask').unsqueeze(1), on_dev('img.mask_valid').unsqueeze(1), normal, norm_mask, world_mat, pix,
                want_normal)
        sync_free = (self.sync_free and is_per_ray_noise(noise) and not eval_mode and self.rendering_technique == 'unisurf'
                     and hasattr(self.model, '_unisurf_sync_free') and pix.is_cuda)
        if hasattr(self.model, 'sync_free'):
            self.model.sync_free = sync_free
        # mask counts: on the device in the sync-free path (Loss divides by them there), else on the host BEFORE the
        # network call (so that no synchronisation sits between forward and backward)
        norm_count = int(norm_mask_gt.sum()) if (normal_gt is not None and not sync_free) else None
        valid_count = int(mask_valid.sum()) if (self.mask_loss and not sync_free) else None
        out = self.model(pix, camera_mat, world_mat, scale_mat, self.rendering_technique, it=it, eval_=eval_mode,
                         noise=noise)
        mask_pred = out.get('acc_map')
        if not self.mask_loss:
            mask_gt = None
        return self.loss(out, rgb_gt, normal_gt, norm_mask_gt, mask_pred, mask_gt, mask_valid,
                         norm_count=norm_count, valid_count=valid_count)
