"""Oracle (CPU restatement) of the stage-2 hot path: SVBRDF / normal /
visibility MLPs, spherical-Gaussian (and GGX microfacet) shading, losses and
the train-step body.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by
tests/golden/stage2_*.npz (tools/gen_golden.py).  Random draws (xyz jitter,
normal jitter) can be injected through ``noise=``; otherwise they are drawn
from torch's global RNG in the reference's order (normal jitter, then xyz
jitter).  Citations are relative to /root/reference/stage2.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def embed(x, n_freqs):
    """NeRF embedding [x, sin(2^k x), cos(2^k x)]_{k<n} (model/embedder.py:6-54)."""
    out = [x]
    for f in (2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)):
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def camera_rays(uv, pose, intrinsics):
    """utils/rend_util.py:90-147 for the 4x4 pose case (device-agnostic)."""
    fx, fy = intrinsics[:, 0, 0], intrinsics[:, 1, 1]
    cx, cy = intrinsics[:, 0, 2], intrinsics[:, 1, 2]
    z = torch.ones_like(uv[:, :, 0])
    x = (uv[:, :, 0] - cx.unsqueeze(-1)) / fx.unsqueeze(-1) * z
    y = (uv[:, :, 1] - cy.unsqueeze(-1)) / fy.unsqueeze(-1) * z
    cam = torch.stack((x, y, z), dim=-1)
    d = torch.einsum('bij,bnj->bni', pose[:, :3, :3], cam)
    return F.normalize(d, dim=2), pose[:, :3, 3]


class MLP(nn.Module):
    """model/renderer.py:17-49.  ReLU stack; the skip concatenates the INPUT
    after layer ``skip_at``; ``final`` is 'sigmoid' (Network) or 'linear'
    (Normal_Network).  state_dict keys: linears.{i}.{weight,bias}."""

    def __init__(self, din, dout, W, depth, skip_at=(), final='linear'):
        super().__init__()
        self.linears = nn.ModuleList(
            [nn.Linear(din, W)] +
            [nn.Linear(W + din if i in skip_at else W, W) for i in range(depth - 1)] +
            [nn.Linear(W, dout)])
        self.skip_at = list(skip_at)
        self.final = final

    def forward(self, x):
        y = x
        last = len(self.linears) - 1
        for li, lyr in enumerate(self.linears):
            y = lyr(y)
            if li != last:
                y = F.relu(y)
            elif self.final == 'sigmoid':
                y = torch.sigmoid(y)
            if li in self.skip_at:
                y = torch.cat([y, x], -1)
        return y


class SGBasis(nn.Module):
    """model/sgbasis.py:7-32."""

    def __init__(self, nbasis=9, specular_rgb=False):
        super().__init__()
        self.nbasis, self.specular_rgb = nbasis, specular_rgb
        self.lobe = nn.Parameter(torch.tensor([np.exp(i) for i in range(2, 11)], dtype=torch.float32),
                                 requires_grad=False)

    def forward(self, v, n, l, albedo, weights):
        h = F.normalize(l + v, dim=-1)
        D = torch.exp(self.lobe[None].clamp(min=0) * ((h * n).sum(-1, keepdim=True) - 1))
        if self.specular_rgb:
            spec = (weights.view(-1, 3, self.nbasis) * D[:, None]).sum(-1).clamp(min=0.0)
        else:
            spec = (weights * D).sum(-1, keepdim=True).clamp(min=0.0)
        return albedo + spec.expand_as(albedo), spec


def _div_no_nan(x, y):
    """model/microfacet.py:20-24."""
    a = torch.div(x, y + 1e-6)
    a = torch.where(torch.isinf(a) | torch.isnan(a), torch.zeros_like(a), a)
    return a


def microfacet_brdf(pts2l, pts2c, normal, albedo, rough, f0=0.05):
    """GGX D*G*F / (4 |l.n| |v.n|) + albedo/pi (model/microfacet.py:35-114).
    pts2l [N,L,3], pts2c [N,3], normal [N,3], albedo [N,3], rough [N,1] -> [N,L,3]."""
    pts2l = F.normalize(pts2l, dim=2, eps=1e-6)
    pts2c = F.normalize(pts2c, dim=1, eps=1e-6)
    normal = F.normalize(normal, dim=1, eps=1e-6)
    h = F.normalize(pts2l + pts2c[:, None, :], dim=2, eps=1e-6)
    fr = f0 + (1 - f0) * (1 - torch.einsum('ijk,ijk->ij', pts2l, h)) ** 5
    alpha = rough ** 2
    # D
    cm = torch.einsum('ijk,ik->ij', h, normal)
    chi = torch.where(cm > 0, 1.0, 0.0)
    cm2 = torch.square(cm)
    tan2 = _div_no_nan(1 - cm2, cm2)
    d = _div_no_nan(alpha ** 2 * chi, np.pi * torch.square(cm2) * torch.square(alpha ** 2 + tan2))
    # G
    cv = torch.einsum('ij,ij->i', normal, pts2c)
    chi_g = torch.where(_div_no_nan(torch.einsum('ijk,ik->ij', h, pts2c), cv[:, None]) > 0, 1.0, 0.0)
    cv2 = torch.clamp(torch.square(cv), 0.0, 1.0)
    tanv2 = torch.clamp(_div_no_nan(1 - cv2, cv2), 0.0, np.inf)
    g = _div_no_nan(chi_g * 2, 1 + torch.sqrt(1 + alpha ** 2 * tanv2[:, None]))
    ln = torch.einsum('ijk,ik->ij', pts2l, normal)
    mf = _div_no_nan(fr * g * d, 4 * torch.abs(ln) * torch.abs(cv)[:, None])
    return mf[:, :, None].repeat(1, 1, 3) + (albedo / np.pi)[:, None, :].expand(-1, mf.shape[1], -1)


class Conf(dict):
    """Duck-typed stand-in for the pyhocon tree the reference reads
    (get_string/get_int/get_float/get_bool with dotted keys)."""

    _MISSING = object()

    def _get(self, key, default):
        node = self
        for part in key.split('.'):
            if not isinstance(node, dict) or part not in node:
                if default is Conf._MISSING:
                    raise KeyError(key)
                return default
            node = node[part]
        return node

    def get_string(self, key, default=_MISSING):
        return self._get(key, default)

    def get_int(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if v is None else int(v)

    def get_float(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if v is None else float(v)

    def get_bool(self, key, default=_MISSING):
        return bool(self._get(key, default))


def bear_conf(**overrides):
    """The hot-path subset of confs/bear.conf:11-99 as a nested dict."""
    c = {
        'train': dict(render_model='sgbasis', nbasis=9, specular_rgb=True, visibility=True,
                      vis_loss=True, light_vis_detach=True, vis_rgb_detach=True, normal_mlp=True,
                      normal_joint=True, shape_pregen=True, light_train=True, light_inten_train=True, light_decay=True),  # (bear.conf:13,17,21; light_inten_train is absent in bunny / armadillo.conf)
        'brdf': dict(net=dict(n_freqs_xyz=10, mlp_width=128, mlp_depth=4, mlp_skip_at=2, xyz_jitter_std=0.01),
                     sgnet=dict(mlp_width=64, mlp_depth=2, mlp_skip_at=-1),
                     fresnel_f0=0.05, light_intensity=2.0),
        'normal': dict(net=dict(n_freqs_xyz=10, mlp_width=128, mlp_depth=4, mlp_skip_at=2, xyz_jitter_std=0.0)),
        'visibility': dict(net=dict(n_freqs_xyz=10, mlp_width=256, mlp_depth=8, mlp_skip_at=4)),
    }
    for k, v in overrides.items():
        node = c
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return Conf(c)


class PSNetwork(nn.Module):
    """model/renderer.py:52-266."""

    def __init__(self, conf):
        super().__init__()
        self.conf = conf
        self.render_model = conf.get_string('train.render_model', default='sgbasis')
        if self.render_model == 'microfacet':
            self.f0 = conf.get_float('brdf.fresnel_f0', default=0.05)
        else:
            nbasis = conf.get_int('train.nbasis', default=9)
            self.specular_rgb = conf.get_bool('train.specular_rgb', default=False)
            self.sgbasis = SGBasis(nbasis=nbasis, specular_rgb=self.specular_rgb)
        self.n_freqs = conf.get_int('brdf.net.n_freqs_xyz')
        dim_emb = 3 + 6 * self.n_freqs
        W, depth = conf.get_int('brdf.net.mlp_width'), conf.get_int('brdf.net.mlp_depth')
        skip = conf.get_int('brdf.net.mlp_skip_at')
        self.albedo_net = MLP(dim_emb, 3, W, depth, skip_at=[skip], final='sigmoid')
        if self.render_model == 'microfacet':
            self.rough_net = MLP(dim_emb, 1, W, depth, skip_at=[skip], final='sigmoid')
        else:
            if self.specular_rgb:
                nbasis *= 3
            self.rough_net = MLP(dim_emb, nbasis, conf.get_int('brdf.sgnet.mlp_width', 128),
                                 conf.get_int('brdf.sgnet.mlp_depth', 4),
                                 skip_at=[conf.get_int('brdf.sgnet.mlp_skip_at', 2)])
            self.nbasis = nbasis
        self.light_int = conf.get_float('brdf.light_intensity', default=4.0)
        self.shape_pregen = conf.get_bool('train.shape_pregen', default=False)
        self.xyz_jitter_std = conf.get_float('brdf.net.xyz_jitter_std', default=0)
        self.normal_mlp = conf.get_bool('train.normal_mlp', default=False)
        if self.normal_mlp:
            self.n_freqs_n = conf.get_int('normal.net.n_freqs_xyz')
            self.normal_net = MLP(3 + 6 * self.n_freqs_n, 3, conf.get_int('normal.net.mlp_width'),
                                  conf.get_int('normal.net.mlp_depth'),
                                  skip_at=[conf.get_int('normal.net.mlp_skip_at')])
            self.normal_joint = conf.get_bool('train.normal_joint', default=False)
            self.normal_jitter_std = conf.get_float('normal.net.xyz_jitter_std', default=0)
            if not self.normal_joint:
                self.normal_net = self.normal_net.eval().requires_grad_(False)
                self.normal_jitter_std = 0
        self.visibility = conf.get_bool('train.visibility', default=False)
        self.light_vis_detach = conf.get_bool('train.light_vis_detach', default=False)
        if self.visibility:
            self.visibility_net = MLP(dim_emb * 2, 1, conf.get_int('visibility.net.mlp_width'),
                                      conf.get_int('visibility.net.mlp_depth'),
                                      skip_at=[conf.get_int('visibility.net.mlp_skip_at')])

    def forward(self, input, albedo_new=None, basis_new=None, noise=None):
        noise = noise or {}
        uv, pose, intr = input['uv'], input['pose'], input['intrinsics']
        object_mask = input['object_mask']
        ray_dirs, _ = camera_rays(uv, pose, intr)
        surface_mask, points, normals = input['surface_mask'], input['points'], input['normal']
        surf = points[surface_mask]
        ns = surf.shape[0]

        out_n = {}
        if self.normal_mlp:  # renderer.py:127-143
            normal_pred = torch.ones_like(points)
            if ns > 0:
                normal_pred[surface_mask] = F.normalize(self.normal_net(embed(surf, self.n_freqs_n)), dim=-1)
                if self.normal_jitter_std > 0:
                    nz = noise.get('normal')
                    if nz is None:
                        nz = torch.normal(0, torch.ones_like(surf) * self.normal_jitter_std)
                    nj = torch.ones_like(points)
                    nj[surface_mask] = F.normalize(self.normal_net(embed(surf + nz, self.n_freqs_n)), dim=-1)
                    out_n['normal_jitter'] = nj
            out_n['normal_pred'] = normal_pred

        sg = self.render_model == 'sgbasis'
        lnum = input['light_direction'].shape[0]
        rgb_values = torch.ones_like(points).repeat(lnum, 1, 1) if lnum > 1 else torch.ones_like(points)
        albedo_values = torch.ones_like(points)
        rough_values = rgb_values.clone() if sg else torch.ones_like(points)
        weight_values = torch.zeros(*points.shape[:-1], self.nbasis) if sg else None
        vis_values = rgb_values.clone()
        jitter = None
        if ns > 0:
            normal = normals[surface_mask] if not self.normal_mlp else normal_pred[surface_mask]
            pts2c = -ray_dirs[surface_mask]
            mask_l = surface_mask.expand(lnum, -1)
            pts2l = input['light_direction'][:, None].expand(rgb_values.shape)[mask_l]  # light-major [L*Ns,3]
            point_emb = embed(surf, self.n_freqs)
            albedo = self.albedo_net(point_emb)
            if albedo_new is not None:
                albedo = torch.from_numpy(albedo_new)[None].expand_as(albedo)
            rough = self.rough_net(point_emb)
            if sg:
                weights = F.relu(rough)
                if basis_new is not None:
                    wn = torch.zeros_like(weights)
                    if self.specular_rgb:
                        wn.view(-1, 3, self.nbasis // 3)[:, :, basis_new] = 2 ** basis_new / 100
                    else:
                        wn.view(-1, 1, self.nbasis)[:, :, basis_new] = 2 ** basis_new / 100
                    weights = wn.reshape(-1, self.nbasis)
                if lnum > 1:
                    brdf, rough = self.sgbasis(l=pts2l, v=pts2c.tile(lnum, 1), n=normal.tile(lnum, 1),
                                               albedo=albedo.tile(lnum, 1), weights=weights.tile(lnum, 1))
                else:
                    brdf, rough = self.sgbasis(l=pts2l, v=pts2c, n=normal, albedo=albedo, weights=weights)
                weight_values[surface_mask] = weights
            else:
                brdf = microfacet_brdf(pts2l.view(lnum, -1, 3).permute(1, 0, 2), pts2c, normal,
                                       albedo, rough, f0=self.f0).permute(1, 0, 2).reshape(-1, 3)
            cos = torch.einsum('lni,ni->ln', pts2l.view(lnum, -1, 3), normal).reshape(-1, 1)  # unclamped
            light_int = input.get('light_intensity', self.light_int)
            if torch.is_tensor(light_int) and light_int.shape[0] > 1:
                light_int = light_int.repeat_interleave(ns, dim=0)
            if self.visibility:
                l_in = pts2l.detach() if self.light_vis_detach else pts2l
                vis = self.visibility_net(torch.cat([point_emb.tile(lnum, 1), embed(l_in, self.n_freqs)], -1))
                v_rgb = vis.detach() if self.conf.get_bool('train.vis_rgb_detach', default=False) else vis
                rgb = (brdf * light_int * cos * v_rgb.clamp(0, 1)).clamp(0, 1)
                vis_values[mask_l] = vis.expand(rgb.shape)
            else:
                rgb = (brdf * light_int * cos).clamp(0, 1)
            rgb_values[mask_l] = rgb
            albedo_values[surface_mask] = albedo
            if sg:
                rough_values[mask_l] = rough.expand(-1, 3)
            else:
                rough_values[surface_mask] = rough.expand(-1, 3)
            if self.xyz_jitter_std > 0:  # renderer.py:211-231
                nz = noise.get('xyz')
                if nz is None:
                    nz = torch.normal(0, torch.ones_like(surf) * self.xyz_jitter_std)
                emb_j = embed(surf + nz, self.n_freqs)
                aj = torch.ones_like(points)
                aj[surface_mask] = self.albedo_net(emb_j)
                rj_net = self.rough_net(emb_j)
                if sg:
                    rj = torch.ones_like(weight_values)
                    rj[surface_mask] = F.relu(rj_net)
                    r_ori = weight_values
                else:
                    rj = torch.ones_like(points)
                    rj[surface_mask] = rj_net.expand(-1, 3)
                    r_ori = rough_values
                jitter = {'albedo_values': albedo_values, 'albedo_jitter': aj,
                          'rough_values': r_ori, 'rough_jitter': rj}

        out = {
            'points': points, 'object_mask': object_mask, 'network_object_mask': surface_mask,
            'sg_rgb_values': rgb_values, 'normal_values': normals,
            'sg_diffuse_albedo_values': albedo_values, 'sg_specular_rgb_values': rough_values,
        }
        if jitter is not None:
            out.update(jitter)
        if self.normal_mlp:
            out.update(out_n)
        if self.visibility:
            out['visibility'] = vis_values
            if 'vis_train_gt' in input or 'light_vis_train' in input:  # renderer.py:251-262
                lv = input['light_vis_train']
                vnum = lv.shape[0]
                mask_v = surface_mask.expand(vnum, -1)
                p2l = lv[:, None].expand(-1, rgb_values.shape[1], -1)[mask_v]
                vt = torch.ones_like(points).repeat(vnum, 1, 1) if vnum > 1 else torch.ones_like(points)
                l_in = p2l.detach() if self.light_vis_detach else p2l
                vis = self.visibility_net(torch.cat([point_emb.tile(vnum, 1), embed(l_in, self.n_freqs)], -1))
                vt[mask_v] = vis.expand(-1, 3)
                out['vis_train'] = vt
        if sg:
            out['sg_weight'] = weight_values
        return out


class MainLoss(nn.Module):
    """model/loss.py:6-92 (device-agnostic: no hard-coded .cuda())."""

    def __init__(self, sg_rgb_weight, loss_type='L1', albedo_smooth_weight=0, rough_smooth_weight=0,
                 vis_weight=1.0):
        super().__init__()
        self.sg_rgb_weight = sg_rgb_weight
        self.albedo_smooth_weight = albedo_smooth_weight
        self.rough_smooth_weight = rough_smooth_weight
        self.vis_weight = vis_weight
        if loss_type not in ('L1', 'L2'):
            raise Exception('Unknown loss_type!')
        self.img_loss = F.l1_loss if loss_type == 'L1' else F.mse_loss

    def _masked(self, fn, a, b, m, om, tail):
        mask = m & om
        if mask.sum() == 0:
            return torch.tensor(0.0)
        mask = mask.expand(b.shape[0], -1)
        return fn(a[mask].reshape(tail), b[mask].reshape(tail))

    def forward(self, model_outputs, ground_truth, model_input=None):
        m, om = model_outputs['network_object_mask'], model_outputs['object_mask']
        rgb_loss = self._masked(self.img_loss, model_outputs['sg_rgb_values'], ground_truth['rgb'], m, om, (-1, 3))
        loss = self.sg_rgb_weight * rgb_loss
        a_loss = r_loss = None
        if 'albedo_jitter' in model_outputs and self.albedo_smooth_weight > 0:
            mask = (m & om)
            a_loss = torch.tensor(0.0) if mask.sum() == 0 else F.l1_loss(
                model_outputs['albedo_values'][mask.expand(model_outputs['albedo_values'].shape[0], -1)],
                model_outputs['albedo_jitter'][mask.expand(model_outputs['albedo_values'].shape[0], -1)])
            loss = loss + self.albedo_smooth_weight * a_loss
        if 'rough_jitter' in model_outputs and self.rough_smooth_weight > 0:
            mask = (m & om)
            r_loss = torch.tensor(0.0) if mask.sum() == 0 else F.l1_loss(
                model_outputs['rough_values'][mask.expand(model_outputs['rough_values'].shape[0], -1)],
                model_outputs['rough_jitter'][mask.expand(model_outputs['rough_values'].shape[0], -1)])
            loss = loss + self.rough_smooth_weight * r_loss
        terms = {'sg_rgb_loss': rgb_loss, 'albedo_smooth_loss': a_loss, 'rough_smooth_loss': r_loss}
        if 'visibility' in model_input and 'visibility' in model_outputs:
            if 'vis_train_gt' in model_input and 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['vis_train_gt']
            elif 'light_vis_train' in model_input and 'vis_train' in model_outputs:
                v, gt = model_outputs['vis_train'][..., 0], model_input['visibility']
            else:
                v, gt = model_outputs['visibility'][..., 0], model_input['visibility']
            vis_loss = self._masked(self.img_loss, v, gt, m, om, (-1,))
            loss = loss + self.vis_weight * vis_loss
            terms['vis_loss'] = vis_loss
        terms['loss'] = loss
        return terms


class NormalLoss(nn.Module):
    """model/loss.py:96-141."""

    def __init__(self, normal_weight, normal_smooth_weight=0):
        super().__init__()
        self.normal_weight, self.normal_smooth_weight = normal_weight, normal_smooth_weight

    def forward(self, model_outputs):
        gt = F.normalize(model_outputs['normal_values'], dim=-1)
        mask = model_outputs['network_object_mask'] & model_outputs['object_mask']
        if mask.sum() == 0:
            n_loss = torch.tensor(0.0)
        else:
            n_loss = F.mse_loss(model_outputs['normal_pred'][mask].reshape(-1, 3), gt[mask].reshape(-1, 3))
        loss = self.normal_weight * n_loss
        s_loss = None
        if 'normal_jitter' in model_outputs and self.normal_smooth_weight > 0:
            s_loss = torch.tensor(0.0) if mask.sum() == 0 else F.l1_loss(
                model_outputs['normal_pred'][mask], model_outputs['normal_jitter'][mask])
            loss = loss + self.normal_smooth_weight * s_loss
        return {'loss': loss, 'normal_loss': n_loss, 'normal_smooth_loss': s_loss}


class TrainStep(object):
    """The step body of TrainRunner.run (trainer.py:355-410,462-464) and the
    train_fix schedule (trainer.py:485-513), without datasets/checkpoints/plots.

    ``light_para`` [n_lights_total,3] and ``light_inten_para`` [n_lights_total,1]
    are sparse embeddings optimised with SparseAdam (trainer.py:126-168); the intensity
    table exists only under ``train.light_inten_train`` (trainer.py:38,154-163: the synthetic
    objects' configurations leave it out and the model then uses its scalar
    brdf.light_intensity, renderer.py:202)."""

    def __init__(self, model, conf, n_lights_total, light_init, lr=5e-4, light_lr=5e-4, light_inten_lr=1e-3,
                 milestones=(), gamma=0.5, loss_kwargs=None, normal_loss_kwargs=None, vis_plus=None):
        self.model, self.conf = model, conf
        # trainer.py:149 -- light_vis_train = clones of the initial (SDPS-Net) light estimates, never optimised;
        # torch.cat(self.light_vis_train) is indexed with l_slt at :377
        self.light_vis_train_all = light_init.detach().clone()
        # trainer.py:209-214 -- vis_plus tables: dict(light=[per view [P,3]], vis=[per view [P,hw]],
        # view_light=[per view [L_v,3] initial estimates], view_vis=[per view [L_v,hw] dataset.visibility], vnum)
        self.vis_plus = vis_plus
        lk = dict(sg_rgb_weight=1.0, loss_type='L1', albedo_smooth_weight=0.05, rough_smooth_weight=0.01,
                  vis_weight=1)
        lk.update(loss_kwargs or {})
        self.loss = MainLoss(**lk)
        nk = dict(normal_weight=1, normal_smooth_weight=0.05)
        nk.update(normal_loss_kwargs or {})
        self.loss_n = NormalLoss(**nk)
        self.sg_optimizer = torch.optim.Adam(model.parameters(), lr=lr)
        self.sg_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.sg_optimizer, list(milestones), gamma=gamma)
        # the switches of trainer.py:36-50 with the reference's defaults
        self.light_train = conf.get_bool('train.light_train', default=False)      # :36 estimated (trained) lights, or the data set's as given
        self.ana_fixlight = conf.get_bool('train.ana_fixlight', default=False)    # :41 light tables stay frozen after iteration 5000
        self.visibility = conf.get_bool('train.visibility', default=False)        # :49
        self.vis_loss = self.visibility and conf.get_bool('train.vis_loss', default=False)  # :50
        self.normal_train = conf.get_bool('train.normal_mlp', default=False) and conf.get_bool('train.normal_joint', default=False)  # :42-44
        self.light_para = nn.Embedding(n_lights_total, 3, sparse=True)
        self.light_para.weight.data.copy_(light_init)
        if not self.light_train:  # trainer.py:126: no table, no optimiser; kept here as a frozen constant the step never reads
            self.light_para.requires_grad_(False)
        self.light_inten_train = self.light_train and conf.get_bool('train.light_inten_train', default=False)  # trainer.py:38,154
        self.light_decay = conf.get_bool('train.light_decay', default=False)  # trainer.py:40: without it the light tables keep their lr
        self.light_inten_para = nn.Embedding(n_lights_total, 1, sparse=True)
        nn.init.constant_(self.light_inten_para.weight, model.light_int)
        groups = [{'params': list(self.light_para.parameters())}]
        if self.light_inten_train:  # trainer.py:154-163
            groups.append({'params': list(self.light_inten_para.parameters()), 'lr': light_inten_lr})
        else:
            self.light_inten_para.requires_grad_(False)  # (kept as a constant table: never read by the step)
        self.light_optimizer = torch.optim.SparseAdam(groups, lr=light_lr)
        self.light_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.light_optimizer, list(milestones),
                                                                    gamma=gamma)
        self.cur_iter = 0

    def train_fix(self):
        if self.cur_iter == 0:
            self._ori = (self.loss.sg_rgb_weight, self.loss.albedo_smooth_weight,
                         self.loss.rough_smooth_weight, self.loss.vis_weight)
            self.loss.sg_rgb_weight = 0
            self.loss.albedo_smooth_weight = 0
            self.loss.rough_smooth_weight = 0
            self.loss.vis_weight = 10
            self.model.albedo_net.eval().requires_grad_(False)
            self.model.rough_net.eval().requires_grad_(False)
            if self.visibility and not self.vis_loss:  # trainer.py:498-499 (never released again)
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
        if train_order:
            self.train_fix()
        model_input = dict(model_input)
        if self.light_train:  # trainer.py:368-379; otherwise the batch's own 'light_direction' is used as given
            model_input['light_direction'] = F.normalize(self.light_para(l_slt), p=2, dim=-1)
            if self.light_inten_train:  # trainer.py:378-379
                model_input['light_intensity'] = self.light_inten_para(l_slt)
        if self.vis_plus is not None and vidx is not None:
            # trainer.py:377 is overwritten by :384-392 when train.vis_plus is set
            vp = self.vis_plus
            light_plus = torch.cat([vp['light'][vidx], vp['view_light'][vidx]], dim=0)            # :386,388
            vis_plus_v = torch.cat([vp['vis'][vidx].reshape(len(vp['light'][vidx]), -1), vp['view_vis'][vidx]], dim=0)  # :387
            sidx = torch.tensor(np.random.choice(np.arange(len(light_plus)), vp['vnum'], replace=False)).long()  # :389
            model_input['light_vis_train'] = light_plus[sidx]                                       # :390
            assert light_plus.shape[0] == vis_plus_v.shape[0]                                       # :391
            model_input['vis_train_gt'] = vis_plus_v[sidx][:, model_input['sampling_idx'][0]]       # :392
        elif self.light_train and 'light_vis_train' not in model_input:
            model_input['light_vis_train'] = F.normalize(self.light_vis_train_all[l_slt], p=2, dim=-1)  # :377
        out = self.model(model_input, noise=noise)
        terms = self.loss(out, ground_truth, model_input)
        loss = terms['loss']
        if self.normal_train:  # trainer.py:397-399
            terms_n = self.loss_n(out)
            loss = loss + terms_n['loss']
        else:
            terms_n = {'loss': None, 'normal_loss': None, 'normal_smooth_loss': None}
        self.sg_optimizer.zero_grad()
        train_light = self.light_para.weight.requires_grad
        if train_light:
            self.light_optimizer.zero_grad()
        loss.backward()
        self.sg_optimizer.step()
        if train_light:
            self.light_optimizer.step()
        self.cur_iter += 1
        self.sg_scheduler.step()
        if train_light and self.light_decay:  # trainer.py:463-464
            self.light_scheduler.step()
        terms = dict(terms)
        terms['total'] = loss
        terms['normal_loss'] = terms_n['normal_loss']
        return terms, out


def psnr(img1, img2, mask=None):
    """trainer.py:268-276."""
    if mask is not None:
        m = mask.to(torch.bool)[:, None]
        img1, img2 = torch.masked_select(img1, m).view(-1, 3), torch.masked_select(img2, m).view(-1, 3)
    mse = ((img1 - img2) ** 2).mean()
    return 100.0 if mse == 0 else float(-10.0 * torch.log10(mse))
