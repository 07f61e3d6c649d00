"""Stage-1 -> stage-2 on-disk hand-off.

Writes / reads exactly the directory layout stage1/shape_extract.py:148-171 produces and
stage2/datasets/dataset.py:88-127 + stage2/trainer.py:207-214 consume:

    <dir>/points/view_XX.npy      float32 [h, w, 3]
    <dir>/normal/view_XX.npy      float32 [h, w, 3]
    <dir>/mask/view_XX.npy        bool    [h, w]
    <dir>/visibility/view_XX.npy  float32 [L, w, h]  (reshape(L,h,w).transpose(0,2,1), shape_extract.py:157)
    <dir>/vis_plus/view_XX.npy    float32 [P, w, h]  + vis_plus/light_dir.json {"view_XX": [[x,y,z], ...]}

The renderer walks the image x-major (stage1/model/common.py:73 arange_pixels), hence the reference's
``to_hw`` = reshape(w, h, -1).permute(1, 0, 2) (stage1/model/training.py:18), reproduced here.
"""
import json
import os

import numpy as np
import torch


def arange_pixels(h, w, device):
    """Integer pixel grid, x-major, [1, h*w, 2] (stage1/model/common.py:55-93 for batch_size 1)."""
    xs, ys = torch.meshgrid(torch.arange(0, w, device=device), torch.arange(0, h, device=device), indexing='ij')
    return torch.stack([xs, ys], dim=-1).long().view(1, -1, 2)


def to_hw(x, h, w):
    return x.reshape(w, h, -1).permute(1, 0, 2)


def sample_vis_plus_dirs(world_mat, vnum=256, semisphere=False, rnum=10000, rng=None):
    """The extra visibility-supervision directions of a view as stage1/shape_extract.py:117-129 draws them: ``rnum`` isotropic
    unit vectors (np.random.normal, normalised), with ``semisphere`` only those facing the camera ((v . view_dir) < 0, view_dir =
    world_mat[0, :3, 2]), thinned to ``vnum`` by farthest-point sampling with a random start.  The reference calls torch_cluster's
    ``fps`` for the last step; this is the same greedy farthest-first selection on the host (the start index comes from ``rng``,
    so the SET of directions is not bit-pinned to torch_cluster's -- any well-spread set serves: the directions are stored beside
    the maps in vis_plus/light_dir.json).  -> float32 tensor [vnum, 3]."""
    rng = rng if rng is not None else np.random
    view_dir = np.asarray(world_mat[0, :3, 2].detach().cpu() if torch.is_tensor(world_mat) else world_mat[0, :3, 2], dtype=np.float64)
    vec = rng.normal(size=(rnum, 3))
    unit = vec / np.linalg.norm(vec, axis=-1, keepdims=True)
    if semisphere:
        unit = unit[(unit * view_dir).sum(-1) < 0]
    assert unit.shape[0] >= vnum, 'sample_vis_plus_dirs: fewer candidates than directions asked for'
    # (np.random / RandomState draw with randint; a numpy Generator -- np.random.default_rng() -- only has integers)
    chosen = [int((rng.integers if hasattr(rng, 'integers') else rng.randint)(unit.shape[0]))]
    dist = np.full(unit.shape[0], np.inf)
    for _ in range(vnum - 1):
        dist = np.minimum(dist, ((unit - unit[chosen[-1]]) ** 2).sum(-1))
        chosen.append(int(dist.argmax()))
    return torch.from_numpy(unit[chosen]).float()


@torch.no_grad()
def export_view(renderer, camera_mat, world_mat, scale_mat, h, w, out_dir, view_id, light_dir=None, vis_plus_dir=None,
                chunk=32000, it=100000):
    """shape_extract.py:120-171 for one view.  ``light_dir`` [L,3] (view's calibrated/estimated lights);
    ``vis_plus_dir`` [P,3] extra supervision directions."""
    dev = camera_mat.device
    p_loc = arange_pixels(h, w, dev).float()
    lights = light_dir
    n_ori = 0 if light_dir is None else light_dir.shape[0]
    if light_dir is not None and vis_plus_dir is not None:
        lights = torch.cat([light_dir, vis_plus_dir], dim=0)
    masks, normals, points, vis = [], [], [], []
    for px in torch.split(p_loc, chunk, dim=1):
        out = renderer(px, camera_mat, world_mat, scale_mat, 'shape_extract', add_noise=False, eval_=True, it=it,
                       visibility=lights is not None, light_dir=lights)
        masks.append(out['mask'])
        normals.append(out['normal'])
        points.append(out['points'])
        if lights is not None:
            vis.append(out['visibility'])
    name = 'view_{:02d}.npy'.format(view_id)
    for sub in ('points', 'normal', 'mask', 'visibility', 'vis_plus'):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    mask_all = to_hw(torch.cat(masks, dim=1), h, w).cpu().numpy()[..., 0]
    np.save(os.path.join(out_dir, 'points', name), np.ascontiguousarray(to_hw(torch.cat(points, dim=1), h, w).cpu().numpy().astype(np.float32)))
    np.save(os.path.join(out_dir, 'normal', name), np.ascontiguousarray(to_hw(torch.cat(normals, dim=1), h, w).cpu().numpy().astype(np.float32)))
    np.save(os.path.join(out_dir, 'mask', name), np.ascontiguousarray(mask_all.astype(bool)))
    if lights is not None:
        v = torch.cat(vis, dim=1).cpu().numpy()
        # NOTE reference quirk reproduced verbatim (shape_extract.py:157): reshape(L, h, w).transpose(0, 2, 1) on
        # the x-major pixel list.  It equals the correct un-flattening only for square images.
        np.save(os.path.join(out_dir, 'visibility', name),
                np.ascontiguousarray(v[:n_ori].reshape(n_ori, h, w).transpose(0, 2, 1).astype(np.float32)))
        if vis_plus_dir is not None:
            n_plus = vis_plus_dir.shape[0]
            np.save(os.path.join(out_dir, 'vis_plus', name),
                    np.ascontiguousarray(v[n_ori:].reshape(n_plus, h, w).transpose(0, 2, 1).astype(np.float32)))
            jpath = os.path.join(out_dir, 'vis_plus', 'light_dir.json')
            table = json.load(open(jpath)) if os.path.exists(jpath) else {}
            table['view_{:02d}'.format(view_id)] = vis_plus_dir.cpu().numpy().astype(np.float32).tolist()
            with open(jpath, 'w') as f:
                json.dump(table, f, indent=4)
    return mask_all


def load_view(shape_dir, view_id, device='cpu', with_visibility=True):
    """The per-view tensors stage2/datasets/dataset.py:100-127 builds: points [1,hw,3], normal [1,hw,3],
    surface_mask [1,hw] (row-major h*w flattening), visibility [L,hw], vis_plus [P,hw] + its directions."""
    name = 'view_{:02d}.npy'.format(view_id)
    pts = np.load(os.path.join(shape_dir, 'points', name))
    out = {
        'points': torch.from_numpy(pts.astype(np.float32)).reshape(1, -1, 3).to(device),
        'normal': torch.from_numpy(np.load(os.path.join(shape_dir, 'normal', name)).astype(np.float32)).reshape(1, -1, 3).to(device),
        'surface_mask': torch.from_numpy(np.load(os.path.join(shape_dir, 'mask', name))).reshape(1, -1).to(device),
        'img_res': list(pts.shape[:2]),
    }
    vpath = os.path.join(shape_dir, 'visibility', name)
    if with_visibility and os.path.exists(vpath):
        v = torch.from_numpy(np.load(vpath)).float()
        out['visibility'] = v.reshape(v.shape[0], -1).to(device)
    ppath = os.path.join(shape_dir, 'vis_plus', name)
    if with_visibility and os.path.exists(ppath):
        v = torch.from_numpy(np.load(ppath)).float()
        out['vis_plus'] = v.reshape(v.shape[0], -1).to(device)
        table = json.load(open(os.path.join(shape_dir, 'vis_plus', 'light_dir.json')))
        out['vis_plus_light'] = torch.tensor(np.array(table['view_{:02d}'.format(view_id)], dtype=np.float32)).to(device)
    return out


class ViewSampler(object):
    """The per-item sampling of stage2/datasets/dataset.py:137-199 (multi_light layout) over views that are already
    in memory: ``views[v]`` = load_view(...) dictionaries, ``images[v]`` [L_v, h*w, 3], ``object_masks[v]`` [h*w] bool,
    ``light_direction[v]`` [L_v, 3], ``poses[v]`` [4,4], ``intrinsics`` [4,4].  Image decoding (PNG, imageio) stays
    outside: this is the wire format between the two accelerated stages, not the file I/O.

    Per item: a fresh subset of ``light_bs`` lights (np.random.choice without replacement, dataset.py:149-151) and,
    when ``n_pixels`` is set, ``n_pixels`` in-mask pixels (dataset.py:182-195).  The draws use the global
    ``np.random`` stream in the reference's order (lights, then pixels) unless ``rng`` is given, so a seeded run
    samples exactly what the reference samples."""

    def __init__(self, views, images, object_masks, light_direction, poses, intrinsics, light_bs, n_pixels=None,
                 sample_in_mask=True, vis_loss=True, split='train', gt_normal=None, rng=None):
        self.views, self.images, self.object_masks = views, images, object_masks
        self.light_direction, self.poses, self.intrinsics = light_direction, poses, intrinsics
        self.light_bs, self.n_pixels, self.sample_in_mask = light_bs, n_pixels, sample_in_mask
        self.vis_loss, self.split, self.gt_normal = vis_loss, split, gt_normal
        self.rng = rng if rng is not None else np.random
        self.img_res = views[0]['img_res']
        self.total_pixels = self.img_res[0] * self.img_res[1]
        self.sampling_idx = None if n_pixels is None else torch.zeros(n_pixels, dtype=torch.long)

    def __len__(self):
        return len(self.views)

    def uv_grid(self):
        h, w = self.img_res
        uv = np.mgrid[0:h, 0:w].astype(np.int32)
        uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float()  # (x, y) order, dataset.py:138-140
        return uv.reshape(2, -1).transpose(1, 0)

    def __getitem__(self, idx):
        v = self.views[idx]
        n_l = self.light_direction[idx].shape[0]
        if self.split == 'train' and n_l >= self.light_bs:
            lidx = torch.tensor(self.rng.choice(np.arange(n_l), self.light_bs, replace=False)).long()
        else:
            lidx = torch.arange(n_l).long()
        omask = self.object_masks[idx]
        uv = self.uv_grid()
        sample = {'object_mask': omask, 'uv': uv, 'vidx': torch.tensor(idx), 'intrinsics': self.intrinsics, 'lidx': lidx,
                  'normal': v['normal'][0], 'points': v['points'][0], 'surface_mask': v['surface_mask'][0],
                  'light_direction': self.light_direction[idx][lidx]}
        if self.gt_normal is not None:
            sample['gt_normal'] = self.gt_normal[idx]
        if self.vis_loss:
            sample['visibility'] = v['visibility'][lidx]
        img = self.images[idx][lidx] * omask[None, :, None]
        ground_truth = {'rgb': img}
        if self.sampling_idx is not None:
            if self.sample_in_mask:
                pick = np.arange(self.total_pixels)[omask.numpy()]
                self.sampling_idx = torch.tensor(self.rng.choice(pick, min(self.sampling_idx.shape[0], pick.shape[0]),
                                                                 replace=False)).long()
            s = self.sampling_idx
            ground_truth['rgb'] = img[:, s, :]
            sample['object_mask'] = omask[s]
            sample['uv'] = uv[s, :]
            for k in ('normal', 'points'):
                sample[k] = sample[k][s, :]
            sample['surface_mask'] = sample['surface_mask'][s]
            if self.vis_loss:
                sample['visibility'] = sample['visibility'][:, s]
            sample['sampling_idx'] = s.clone()
        sample['pose'] = self.poses[idx]
        return idx, sample, ground_truth

    def batch(self, idx, device=None):
        """(vidx, model_input, ground_truth, l_slt) as TrainRunner.run sees them after DataLoader(batch_size=1) collation
        and its multi_light un-batching (stage2/trainer.py:364-379): per-pixel tensors carry a leading batch dimension of
        1, the per-light ones (rgb [L,n,3], light_direction [L,3], visibility [L,n]) do not, and ``l_slt`` = the rows of
        the concatenated per-view light tables (sum of the preceding views' light counts + lidx)."""
        _, sample, gt = self[idx]
        if 'visibility' not in sample:
            # trainer.py:366 strips the batch dimension of model_input['visibility'] under multi_light unconditionally: a data set
            # built without train.vis_loss (dataset.py:168-169) fails there with this KeyError
            raise KeyError('visibility')
        mi = {}
        for k, v in sample.items():
            if k in ('light_direction', 'visibility', 'lidx'):
                mi[k] = v
            elif torch.is_tensor(v):
                mi[k] = v[None]
            else:
                mi[k] = v
        # index list of the batch's surface pixels, built here on the host (PSNetwork.forward then needs no nonzero())
        mi['surface_idx'] = sample['surface_mask'].nonzero(as_tuple=True)[0]
        accu = [int(ld.shape[0]) for ld in self.light_direction]
        l_slt = sum(accu[:idx]) + sample['lidx']
        gt = dict(gt)
        if device is not None:
            mi = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in mi.items()}
            gt = {k: v.to(device) for k, v in gt.items()}
            l_slt = l_slt.to(device)
        return idx, mi, gt, l_slt
