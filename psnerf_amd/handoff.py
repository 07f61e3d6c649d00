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
def extract_view(renderer, camera_mat, world_mat, scale_mat, h, w, light_dir=None, vis_plus_dir=None, chunk=32000, it=100000):
    """The extraction loop of shape_extract.py:120-147 for one view, before anything is written: the renderer's 'shape_extract'
    over the x-major pixel list of the h x w image in chunks, shadow-ray visibility towards ``light_dir`` [L,3] followed by
    ``vis_plus_dir`` [P,3].  Returns {'mask' [1, hw], 'normal' [1, hw, 3], 'points' [1, hw, 3], 'visibility' [L + P, hw] or None,
    'pixels' [1, hw, 2]} on the renderer's device (pixel order = arange_pixels)."""
    dev = camera_mat.device
    p_loc = arange_pixels(h, w, dev).float()
    lights = light_dir
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
    return {'mask': torch.cat(masks, dim=1), 'normal': torch.cat(normals, dim=1), 'points': torch.cat(points, dim=1),
            'visibility': torch.cat(vis, dim=1) if lights is not None else None, 'pixels': p_loc}


def export_view(renderer, camera_mat, world_mat, scale_mat, h, w, out_dir, view_id, light_dir=None, vis_plus_dir=None,
                chunk=32000, it=100000, extracted=None):
    """shape_extract.py:120-171 for one view.  ``light_dir`` [L,3] (view's calibrated/estimated lights);
    ``vis_plus_dir`` [P,3] extra supervision directions.  ``extracted``: the result of extract_view for these arguments (a caller
    that has it already), else it is computed here."""
    lights = light_dir
    n_ori = 0 if light_dir is None else light_dir.shape[0]
    if light_dir is not None and vis_plus_dir is not None:
        lights = torch.cat([light_dir, vis_plus_dir], dim=0)
    ex = extracted if extracted is not None else extract_view(renderer, camera_mat, world_mat, scale_mat, h, w, light_dir, vis_plus_dir, chunk, it)
    masks, normals, points, vis = [ex['mask']], [ex['normal']], [ex['points']], [ex['visibility']]
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

    def change_sampling_idx(self, sampling_size):
        """dataset.py:219-226: the per-epoch pixel subset when train.sample_in_mask is off (python's ``random.sample``; with
        sample_in_mask every item draws its own subset and this one is only a size)."""
        import random
        if sampling_size == -1:
            self.sampling_idx = None
        else:
            self.sampling_idx = torch.tensor(random.sample(range(self.total_pixels), sampling_size)).long()


class _Draw(object):
    """The host-side random draws of one item, in the reference's order (lights, pixels, vis_plus rows)."""
    __slots__ = ('idx', 'lidx', 'pix', 'vis_rows')

    def __init__(self, idx, lidx, pix, vis_rows):
        self.idx, self.lidx, self.pix, self.vis_rows = idx, lidx, pix, vis_rows


class DeviceViews(object):
    """``ViewSampler`` with the views RESIDENT IN HBM and the batch assembled on the device.

    The reference builds every training item on the host (stage2/datasets/dataset.py:137-199: ``imgs[view][lidx] * mask`` =
    light_bs x h*w x 3 floats materialised per iteration, then every per-pixel tensor indexed with the drawn pixel list) and
    uploads the whole batch (trainer.py:381-382); at BEAR shapes that is 0.3 - 1.8 s of host time per 25 ms GPU step.  Here
    every per-view table -- images (uint8 / uint16 as decoded when every value is k / 255, else float32), object / surface
    masks, stage-1 points, normals, visibility maps, the vis_plus tables of ``VisPlus`` -- is uploaded ONCE (20 views x 96
    lights x 612 x 512: 1.8 GB as uint8, 7.2 GB as float32, of 288 GB), and a step consists of

      * ``draw(idx)``   -- the random draws, on the host, from the same ``np.random`` stream in the reference's order: the light
                           subset (dataset.py:149-151), the in-mask pixel subset (:182-185), the vis_plus rows (trainer.py:389);
      * ``assemble(d)`` -- ONE small host -> device copy of the drawn index lists (pinned, asynchronous) and ONE gather launch
                           (csrc/views.hip psn_view_batch) that writes the batch; under data parallelism (``dp``) a rank gathers only
                           ITS ``slice_bounds`` share of the drawn pixel list, and the index list of its surface pixels comes from
                           the host copy of the view's surface mask (no nonzero(), no synchronisation).

    ``batch(idx)`` = both; the result equals ``ViewSampler.batch(idx, device)`` (``dp.shard_stage2`` of it under data parallelism)
    bit for bit, with 'light_vis_train' / 'vis_train_gt' of the vis_plus draw already in the dictionary when a ``VisPlus`` table
    is attached (``TrainStep.step`` is then called WITHOUT ``vidx``: the draw has been made here).  ``loader(order)`` runs draws and gathers of the next items in a worker thread on a side
    stream, so that the ~7 ms a 280k-pixel ``np.random.choice`` takes (numpy releases the GIL inside it) never sits in front of a step."""

    def __init__(self, views, images, object_masks, light_direction, poses, intrinsics, light_bs, device, n_pixels=None,
                 sample_in_mask=True, vis_loss=True, split='train', gt_normal=None, rng=None, dp=None, vis_plus=None,
                 image_store='auto'):
        from . import hip
        self.hip = hip
        self.device = torch.device(device)
        dev = self.device
        self.light_bs, self.n_pixels, self.sample_in_mask = int(light_bs), n_pixels, sample_in_mask
        self.vis_loss, self.split = vis_loss, split
        self.rng = rng if rng is not None else np.random
        self.dp, self.vis_plus = dp, vis_plus
        self.img_res = list(views[0]['img_res'])
        self.total_pixels = self.img_res[0] * self.img_res[1]
        hw = self.total_pixels
        self.sampling_idx = None if n_pixels is None else torch.zeros(n_pixels, dtype=torch.long)
        self.n_lights = [int(ld.shape[0]) for ld in light_direction]
        self.light_offset = np.concatenate([[0], np.cumsum(self.n_lights)]).astype(np.int64)
        # value tables of integer images: (float32) k / 255. exactly as numpy forms it in dataset.py:121
        self._lut = {}
        self.tables, self.pick, self.surf_host = [], [], []
        for v, view in enumerate(views):
            img = images[v]
            img = torch.as_tensor(img) if not torch.is_tensor(img) else img
            img = self._store(img.reshape(img.shape[0], hw, 3), image_store)
            om = torch.as_tensor(object_masks[v]).reshape(hw).bool()
            sm = view['surface_mask'].reshape(hw).bool()
            t = {'images': img.to(dev).contiguous(), 'width': self.img_res[1],
                 'object_mask': om.to(dev).contiguous(), 'surface_mask': sm.to(dev).contiguous(),
                 'points': view['points'].reshape(hw, 3).float().to(dev).contiguous(),
                 'normal': view['normal'].reshape(hw, 3).float().to(dev).contiguous(),
                 'light_direction': torch.as_tensor(light_direction[v]).float().to(dev).contiguous()}
            if img.dtype != torch.float32:
                t['lut'] = self._value_table(img.dtype)
            if vis_loss:
                t['visibility'] = view['visibility'].reshape(-1, hw).float().to(dev).contiguous()
            if vis_plus is not None:
                t['vis_plus'] = vis_plus.vis[v].contiguous()
                t['vis_plus_light'] = vis_plus.lights[v].contiguous()
            self.tables.append(t)
            self.pick.append(np.arange(hw)[om.cpu().numpy()])  # dataset.py:184 (fixed per view: cached)
            self.surf_host.append(sm.cpu().numpy())
        self.poses = [torch.as_tensor(p).float().reshape(1, 4, 4).to(dev) for p in poses]
        self.intrinsics = torch.as_tensor(intrinsics).float().reshape(1, *torch.as_tensor(intrinsics).shape[-2:]).to(dev)
        self.vidx_dev = [torch.tensor(v, device=dev) for v in range(len(views))]
        self.gt_normal = None if gt_normal is None else [torch.as_tensor(g).float().reshape(1, hw, 3).to(dev) for g in gt_normal]
        self._pinned = {}   # staging rows of the index lists (one ring per thread)
        self.host_seconds = {'draw': 0.0, 'assemble': 0.0, 'items': 0}

    # ---- storage ----------------------------------------------------------------------------------------------------------
    def _value_table(self, dtype):
        if dtype not in self._lut:
            n = 256 if dtype == torch.uint8 else 65536
            self._lut[dtype] = torch.from_numpy(np.arange(n).astype(np.float32) / 255.).to(self.device)
        return self._lut[dtype]

    @staticmethod
    def _store(img, image_store):
        """Integer images stay as decoded; float images are stored as uint8 when EVERY value is exactly (float32) k / 255. (what
        dataset.py:121 makes of an 8-bit PNG) and image_store allows it -- the gathered batch is bit-identical either way."""
        if img.dtype in (torch.uint8, torch.uint16):
            return img
        img = img.float()
        if image_store == 'float32':
            return img
        lut = torch.from_numpy(np.arange(256).astype(np.float32) / 255.)
        k = (img * 255.0).round().clamp_(0, 255).to(torch.uint8)
        if bool((lut[k.long()] == img).all()):
            return k
        assert image_store == 'auto', 'DeviceViews: the images are not 8-bit values / 255 (image_store=%r)' % (image_store,)
        return img

    def __len__(self):
        return len(self.tables)

    def change_sampling_idx(self, sampling_size):
        import random
        if sampling_size == -1:
            self.sampling_idx = None
        else:
            self.sampling_idx = torch.tensor(random.sample(range(self.total_pixels), sampling_size)).long()

    def resident_bytes(self):
        return sum(t.numel() * t.element_size() for tab in self.tables for t in tab.values() if torch.is_tensor(t))

    # ---- host half: the draws ----------------------------------------------------------------------------------------------
    def draw(self, idx):
        import time
        t0 = time.thread_time()   # CPU time of the calling thread: waits (slots, events) are back-pressure, not host work
        idx = int(idx)
        n_l = self.n_lights[idx]
        if self.split == 'train' and n_l >= self.light_bs:
            lidx = self.rng.choice(np.arange(n_l), self.light_bs, replace=False).astype(np.int64)   # dataset.py:149
        else:
            lidx = np.arange(n_l, dtype=np.int64)
        pix = None
        if self.sampling_idx is not None:
            if self.sample_in_mask:
                pick = self.pick[idx]
                pix = self.rng.choice(pick, min(self.sampling_idx.shape[0], pick.shape[0]), replace=False).astype(np.int64)  # :184-185
                self.sampling_idx = torch.from_numpy(pix)
            else:
                pix = self.sampling_idx.numpy().astype(np.int64)
        rows = None
        if self.vis_plus is not None:
            rows = self.rng.choice(np.arange(self.tables[idx]['vis_plus'].shape[0]), self.vis_plus.vnum, replace=False).astype(np.int64)  # trainer.py:389
        self.host_seconds['draw'] += time.thread_time() - t0
        return _Draw(idx, lidx, pix, rows)

    # ---- device half: index upload + one gather launch ----------------------------------------------------------------------
    def _staging(self, n_words):
        import threading
        key = threading.get_ident()
        ring = self._pinned.get(key)
        if ring is None or ring['buf'].shape[1] < n_words:
            ring = self._pinned[key] = {'buf': torch.empty(8, max(n_words, 1024), dtype=torch.int64).pin_memory(), 'ev': [None] * 8, 'turn': 0}
        k = ring['turn'] % 8
        ring['turn'] += 1
        if ring['ev'][k] is not None:
            ring['ev'][k].synchronize()   # the copy that last read this row (8 items ago) has long completed
        return ring, k

    def assemble(self, d, out=None):
        """(idx, model_input, ground_truth, l_slt) of the draw ``d`` on the device; ``out``: preallocated outputs (loader slots)."""
        import time
        t0 = time.thread_time()
        hip, dev, tab = self.hip, self.device, self.tables[d.idx]
        n_all = self.total_pixels if d.pix is None else int(d.pix.shape[0])
        lo, hi = (0, n_all) if (self.dp is None or not self.dp.enabled) else self.dp.slice_bounds(n_all)
        n = hi - lo
        pix = None if d.pix is None else d.pix[lo:hi]
        surf = np.flatnonzero(self.surf_host[d.idx][lo:hi] if pix is None else self.surf_host[d.idx][pix]).astype(np.int64)
        L, V, ns = int(d.lidx.shape[0]), (0 if d.vis_rows is None else int(d.vis_rows.shape[0])), int(surf.shape[0])
        n_pix_words = 0 if pix is None else n
        words = 2 * L + V + n_pix_words + ns
        ring, k = self._staging(words)
        row = ring['buf'][k].numpy()
        row[0:L] = d.lidx
        row[L:2 * L] = self.light_offset[d.idx] + d.lidx        # l_slt: rows of the concatenated light tables (trainer.py:370-373)
        o = 2 * L
        if V:
            row[o:o + V] = d.vis_rows
        o += V
        if pix is not None:
            row[o:o + n] = pix
        row[o + n_pix_words:o + n_pix_words + ns] = surf
        if out is None:
            out = self.alloc_outputs(L, n, V)
        idx_dev = out['_index'][:words]
        idx_dev.copy_(ring['buf'][k, :words], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring['ev'][k] = ev
        lidx_dev, l_slt = idx_dev[0:L], idx_dev[L:2 * L]
        rows_dev = idx_dev[2 * L:2 * L + V] if V else None
        pix_dev = idx_dev[o:o + n] if pix is not None else None
        surf_dev = idx_dev[o + n_pix_words:o + n_pix_words + ns]
        g = {'rgb': out['rgb'][:L * n * 3].view(L, n, 3), 'object_mask': out['object_mask'][:n], 'surface_mask': out['surface_mask'][:n],
             'uv': out['uv'][:n * 2].view(n, 2), 'points': out['points'][:n * 3].view(n, 3), 'normal': out['normal'][:n * 3].view(n, 3),
             'light_direction': out['light_direction'][:L * 3].view(L, 3)}
        if self.vis_loss:
            g['visibility'] = out['visibility'][:L * n].view(L, n)
        if V:
            g['vis_train_gt'] = out['vis_train_gt'][:V * n].view(V, n)
        hip.view_batch(tab, lidx_dev, pix_dev, n, g, pix0=lo, vidx=rows_dev)
        mi = {'object_mask': g['object_mask'][None], 'uv': g['uv'][None], 'vidx': self.vidx_dev[d.idx][None], 'intrinsics': self.intrinsics,
              'lidx': lidx_dev, 'normal': g['normal'][None], 'points': g['points'][None], 'surface_mask': g['surface_mask'][None],
              'light_direction': g['light_direction'], 'pose': self.poses[d.idx], 'surface_idx': surf_dev}
        if self.gt_normal is not None:
            mi['gt_normal'] = self.gt_normal[d.idx]
        if self.vis_loss:
            mi['visibility'] = g['visibility']
        if pix is not None:
            mi['sampling_idx'] = pix_dev[None]
        if V:
            mi['light_vis_train'] = torch.index_select(tab['vis_plus_light'], 0, rows_dev, out=out['light_vis_train'][:V * 3].view(V, 3))  # trainer.py:390
            mi['vis_train_gt'] = g['vis_train_gt']                                              # trainer.py:392
        self.host_seconds['assemble'] += time.thread_time() - t0
        self.host_seconds['items'] += 1
        return d.idx, mi, {'rgb': g['rgb']}, l_slt

    def alloc_outputs(self, L, n, V):
        dev = self.device
        f = lambda m: torch.empty(max(int(m), 1), device=dev, dtype=torch.float32)
        return {'rgb': f(L * n * 3), 'object_mask': torch.empty(max(n, 1), device=dev, dtype=torch.bool),
                'surface_mask': torch.empty(max(n, 1), device=dev, dtype=torch.bool), 'uv': f(n * 2), 'points': f(n * 3), 'normal': f(n * 3),
                'light_direction': f(L * 3), 'visibility': f(L * n if self.vis_loss else 1), 'vis_train_gt': f(V * n), 'light_vis_train': f(V * 3),
                '_index': torch.empty(2 * L + V + 2 * n + 16, device=dev, dtype=torch.int64)}

    def batch(self, idx, device=None):
        """ViewSampler.batch's signature (``device`` is this store's)."""
        assert device is None or torch.device(device) == self.device
        return self.assemble(self.draw(idx))

    # ---- prefetching loader ---------------------------------------------------------------------------------------------------
    def loader(self, order, depth=2):
        """Iterator over ``batch(idx) for idx in order`` whose draws and gathers run up to ``depth`` items AHEAD in a worker thread on
        a side stream (the draws stay in ``order``: one thread, one np.random stream).  The tensors of an item live in one of
        depth + 2 fixed slots and stay valid until the NEXT item is fetched: consume (or copy -- GraphedTrainStep copies into its
        own static buffers) an item before asking for the next."""
        return _Prefetcher(self, order, depth)


class _Prefetcher(object):
    def __init__(self, store, order, depth):
        import queue
        import threading
        self.store, self.order = store, list(int(i) for i in order)
        self.n_slots = int(depth) + 2
        self.side = torch.cuda.Stream(device=store.device)
        self.free, self.ready = queue.Queue(), queue.Queue(maxsize=max(1, int(depth)))
        self.slots = [None] * self.n_slots
        self.released = [None] * self.n_slots   # event recorded by the consumer behind the last step that read slot k
        for k in range(self.n_slots):
            self.free.put(k)
        self.current = None
        self.error = None
        self.stop = False
        self.consumer_wait = 0.0
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        st = self.store
        import queue

        def blocking(fn, *a):   # a queue operation that gives up when the consumer has closed the loader
            while not self.stop:
                try:
                    return fn(*a, timeout=0.2)
                except (queue.Empty, queue.Full):
                    continue
            raise StopIteration
        try:
            torch.cuda.set_device(st.device)   # a new thread starts on device 0
            for idx in self.order:
                if self.stop:
                    return
                d = st.draw(idx)
                k = blocking(self.free.get)
                n_all = st.total_pixels if d.pix is None else int(d.pix.shape[0])
                lo, hi = (0, n_all) if (st.dp is None or not st.dp.enabled) else st.dp.slice_bounds(n_all)
                L, V = int(d.lidx.shape[0]), (0 if d.vis_rows is None else int(d.vis_rows.shape[0]))
                need = (L, hi - lo, V)
                with torch.cuda.stream(self.side):
                    if self.slots[k] is None or self.slots[k][0] != need:
                        self.slots[k] = (need, st.alloc_outputs(*need))
                    if self.released[k] is not None:
                        self.side.wait_event(self.released[k])
                    item = st.assemble(d, out=self.slots[k][1])
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                blocking(self.ready.put, (k, ev, item))
            blocking(self.ready.put, None)
        except StopIteration:
            return
        except BaseException as e:  # noqa: BLE001 (handed to the consumer)
            # the error is set BEFORE the sentinel is offered and the consumer also polls it (and the thread's liveness) while it
            # waits, so a full queue under a long step cannot hide it; the sentinel is offered until delivered or the loader closed
            self.error = e
            try:
                blocking(self.ready.put, None)
            except BaseException:  # noqa: BLE001 (closed meanwhile)
                pass

    def __iter__(self):
        return self

    def __next__(self):
        import time
        if self.current is not None:   # the step on the previous item has been enqueued: its slot may be rewritten behind it
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.store.device))
            self.released[self.current] = ev
            self.free.put(self.current)
            self.current = None
        import queue
        t0 = time.perf_counter()
        while True:
            try:
                got = self.ready.get(timeout=0.2)
                break
            except queue.Empty:
                # nothing ready: a worker that died (error or not) will never deliver -- do not block forever (under data
                # parallelism the other ranks would be left inside their collectives)
                if not self.thread.is_alive() and self.ready.empty():
                    got = None
                    break
        self.consumer_wait += time.perf_counter() - t0
        if got is None:
            self.stop = True
            if self.error is not None:
                raise self.error
            raise StopIteration
        k, ev, item = got
        torch.cuda.current_stream(self.store.device).wait_event(ev)
        self.current = k
        return item

    def close(self):
        """Stop the worker (a consumer that leaves the loop early: the draws made ahead are lost -- re-seed to resume in step)."""
        self.stop = True
        self.thread.join(timeout=5.0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        self.stop = True
