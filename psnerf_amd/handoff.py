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
