"""Environment-map relighting (forward-only reuse of the stage-2 path): SURVEY 8(f1).

Reference: stage2/eval.py:99-112 (16 x 32 lat-long light grid), :173-231 (light batches x pixel chunks,
sum over lights WITHOUT solid-angle weights, clip), stage2/utils/eval_utils.py:61-99 (gen_light_xyz),
stage2/utils/general.py:23-52 (split_input / merge_output).  Environment maps: arrays ([h, 2h, 3] float), .npy files, or the
reference's Radiance .hdr / OpenEXR .exr files through the readers of envmap_io.py (cv2 / OpenEXR are not dependencies).

On MI355X the 1024-pixel chunking of the reference is unnecessary (activations never reach HBM in the fused
visibility kernel), so ``pixel_chunk`` defaults to the whole image; the helpers keep the reference's
signatures for callers that still split.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import envmap_io


def gen_light_xyz(envmap_h, envmap_w, envmap_radius=1e2):
    """Lat-long light directions and solid angles (eval_utils.py:61-99, sph2cart :255-296)."""
    lat_step = np.pi / (envmap_h + 2)
    lng_step = 2 * np.pi / (envmap_w + 2)
    lats = np.linspace(np.pi / 2 - lat_step, -np.pi / 2 + lat_step, envmap_h)
    lngs = np.linspace(np.pi - lng_step, -np.pi + lng_step, envmap_w)
    lngs, lats = np.meshgrid(lngs, lats)
    r = envmap_radius * np.ones_like(lats)
    xyz = np.stack((r * np.cos(lats) * np.cos(lngs), r * np.cos(lats) * np.sin(lngs), r * np.sin(lats)), axis=-1)
    sin_colat = np.sin(np.pi / 2 - lats)
    areas = 4 * np.pi * sin_colat / np.sum(sin_colat)
    return xyz.reshape(envmap_h, envmap_w, 3), areas


def load_light(path_or_array, light_h=None):
    """[h, 2h, 3] float32 RGB environment map from an array, a .npy file or the reference's own formats -- Radiance .hdr and
    OpenEXR .exr (eval_utils.py:11-38; decoded by stage2/envmap_io.py, no cv2) --, resized to light_h rows with the bilinear
    half-pixel rule of cv2.resize(..., INTER_LINEAR) (eval_utils.py:19)."""
    if isinstance(path_or_array, str):
        ext = os.path.basename(path_or_array).split('.')[-1].lower()
        if ext == 'npy':
            arr = np.load(path_or_array)
        elif ext == 'exr':
            arr = envmap_io.read_exr(path_or_array)
        elif ext == 'hdr':
            arr = envmap_io.read_hdr(path_or_array)
        else:
            raise NotImplementedError(ext)  # eval_utils.py:17
    else:
        arr = np.asarray(path_or_array)
    arr = arr.astype(np.float32)
    if light_h and tuple(arr.shape[:2]) != (light_h, 2 * light_h):  # eval_utils.py:19 resizes whenever light_h is given
        t = torch.from_numpy(arr).permute(2, 0, 1)[None]
        arr = F.interpolate(t, size=(light_h, 2 * light_h), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    return arr


PIXEL_KEYS = ('uv', 'object_mask', 'gt_normal', 'normal', 'depth', 'points', 'surface_mask', 'visibility')


def split_input(model_input, total_pixels, n_pixels=1024):
    """general.py:23-37 (device-agnostic)."""
    dev = model_input['uv'].device
    out = []
    for idx in torch.split(torch.arange(total_pixels, device=dev), n_pixels, dim=0):
        data = dict(model_input)
        for k in PIXEL_KEYS:
            if k in model_input:
                data[k] = torch.index_select(model_input[k], 1, idx)
        out.append(data)
    return out


def merge_output(res, total_pixels, batch_size):
    """general.py:39-52."""
    merged = {}
    for k in res[0]:
        if res[0][k] is None:
            continue
        if res[0][k].dim() < 3:
            merged[k] = torch.cat([r[k].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        else:
            merged[k] = torch.cat([r[k].reshape(*r[k].shape[:-2], -1, r[k].shape[-1]) for r in res], -2) \
                .reshape(-1, res[0][k].shape[-1])
    return merged


@torch.no_grad()
def render_envmap(model, model_input, env_light, light_h=16, light_batch=64, pixel_chunk=None, envmap_scale=1.0,
                  visibility=False, precision=None):
    """Relit image of one view: sum over the light_h x 2 light_h environment lights (eval.py:199-218).

    model_input: uv [1,N,2], intrinsics, pose, object_mask, normal, points, surface_mask (no lights).
    env_light: [light_h, 2*light_h, 3].  Returns rgb [N,3] (clipped to [0,1]) and, if requested, the
    light-averaged visibility [N,3].  precision='bf16' evaluates visibility_net on the bf16 MFMA engine for this
    call (PSNetwork.inference_precision; default: the model's current setting, 'fp32')."""
    if precision is not None:
        saved = model.inference_precision
        model.inference_precision = precision
        try:
            return render_envmap(model, model_input, env_light, light_h, light_batch, pixel_chunk, envmap_scale, visibility)
        finally:
            model.inference_precision = saved
    dev = model_input['uv'].device
    n_pix = model_input['uv'].shape[1]
    lxyz, _areas = gen_light_xyz(light_h, 2 * light_h, envmap_radius=1)
    lxyz = torch.from_numpy(lxyz.reshape(-1, 3)).float().to(dev)
    env = torch.as_tensor(np.asarray(env_light, dtype=np.float32).reshape(-1, 3) * envmap_scale, device=dev)
    rgb_sum = torch.zeros(n_pix, 3, device=dev)
    vis_sum = torch.zeros(n_pix, 3, device=dev)
    n_lights = lxyz.shape[0]
    # pixel chunks are cut once and shared by all light batches: the model then computes their light-independent
    # parts (positional encodings, BRDF and normal nets) once per chunk instead of once per (chunk, light batch)
    chunks = [dict(model_input)] if pixel_chunk is None else split_input(model_input, n_pix, pixel_chunk)
    model._eval_cache = {}
    # only the light sum of the shaded colour (and the visibility, if asked for) is consumed: the model then writes no other
    # dense [L, N, C] output and reduces the lights on the surface rows (PSNetwork._eval_outputs)
    fast = hasattr(model, '_eval_outputs')
    if fast:
        model._eval_outputs = {'sg_rgb_light_sum'} | ({'visibility'} if visibility else set())
    try:
        for l0 in range(0, n_lights, light_batch):
            light_direction = F.normalize(lxyz[l0:l0 + light_batch], p=2, dim=-1)
            light_intensity = env[l0:l0 + light_batch].contiguous()
            p0 = 0
            for s in chunks:
                s['light_direction'], s['light_intensity'] = light_direction, light_intensity
                out = model(s)
                n = s['uv'].shape[1]
                if fast:
                    rows_sum, idx, const = out['sg_rgb_light_sum']
                    batch_sum = torch.full((n, 3), const, device=dev)  # what out['sg_rgb_values'].sum(0) holds: the fill summed
                    if rows_sum is not None:                            # over the batch's lights, the row sums at the surface pixels
                        batch_sum[idx] = rows_sum
                    rgb_sum[p0:p0 + n] += batch_sum
                else:
                    rgb_sum[p0:p0 + n] += out['sg_rgb_values'].reshape(-1, n, 3).sum(0)
                if visibility:
                    vis = out.get('visibility')
                    vis_sum[p0:p0 + n] += vis.reshape(-1, n, 3).sum(0) if vis is not None else float(light_direction.shape[0])
                p0 += n
    finally:
        model._eval_cache = None
        if fast:
            model._eval_outputs = None
    rgb = rgb_sum.clamp(0, 1)
    return (rgb, vis_sum / n_lights) if visibility else rgb


def eval_lights(light_direction, lidx, light_para=None, light_inten_para=None, light_offset=0):
    """The lights of one light batch of a test view as evaluate() forms them (eval.py:338-345): the data set's directions of the
    view, or -- for a model trained with ``train.light_train`` on all views (``light_para`` given) -- the optimised table rows
    ``light_offset + lidx`` normalised (+ the optimised intensities when ``light_inten_para`` is given).
    -> (light_direction [l, 3], light_intensity [l, 1] | None)."""
    if light_para is None:
        return light_direction[lidx], None
    w = light_para.weight if hasattr(light_para, 'weight') else light_para
    l_slt = (int(light_offset) + lidx).to(w.device)
    inten = None
    if light_inten_para is not None:
        wi = light_inten_para.weight if hasattr(light_inten_para, 'weight') else light_inten_para
        inten = wi.detach()[l_slt]
    return F.normalize(w.detach()[l_slt], p=2, dim=-1), inten


def edit_material(color=None, basis=None, edit_albedo=True, edit_specular=False, rng=None):
    """The material edit of eval.py:121-139: ``albedo_new`` float32 [3] from a '#rrggbb' colour (each channel / 5 / 255; without a
    colour three np.random draws from range(128), / 255) and ``basis_new`` = the index of the single spherical-Gaussian lobe that
    keeps a weight (2^basis / 100, renderer.py:177-183; without an index one np.random draw from range(9)).
    -> (albedo_new | None, basis_new | None, name) -- ``name`` is the reference's output sub-directory."""
    rng = rng if rng is not None else np.random
    albedo_new, basis_new, name = None, None, ''
    if edit_albedo:
        if color is None:
            albedo_new = rng.choice(range(128), size=3)
            name += '#{:02x}{:02x}{:02x}'.format(*list(albedo_new))
        else:
            albedo_new = np.array([int(color.lstrip('#')[i:i + 2], 16) for i in (0, 2, 4)]).astype(np.float32) / 5.
            name = color
        albedo_new = (albedo_new / 255.).astype(np.float32)
    if edit_specular:
        basis_new = int(rng.choice(range(9))) if basis is None else int(basis)
        name = 'sg%d' % (basis_new + 1) if name == '' else name + '_sg%d' % (basis_new + 1)
    return albedo_new, basis_new, name


@torch.no_grad()
def render_view(model, model_input, light_direction, light_intensity=None, light_batch=64, pixel_chunk=None, albedo_new=None,
                basis_new=None):
    """The test-view render of evaluate() (eval.py:314-417; with ``albedo_new`` / ``basis_new`` the material-edit loop :233-312):
    the view under ITS OWN lights in batches of ``light_batch``, pixels in chunks through split_input / merge_output
    (``pixel_chunk`` None = the whole image in one piece: the fused kernels keep no [N, 256] activations in HBM, the reference
    needs 1024-pixel chunks), and the per-view maps the reference writes to disk, as float32 arrays over the N pixels:

        'rgb' [L, N, 3] clipped (:371)         'rough' [L, N, 3] (sgbasis: the specular colour per light; microfacet: [N, 3], :372-375)
        'mask' [N] bool (:382)                 'normal' [N, 3] (normal_pred with a normal net, else the stage-1 normals) x mask (:394-395)
        'albedo' [N, 3] clipped (:400)         'visibility' [L, N, 3] clipped (:406; only for a model with a visibility net)

    model_input: the item of eval.py:327-337 (uv [1,N,2] x-major grid, intrinsics, pose, object_mask, normal, points, surface_mask).
    light_direction [L, 3] (see eval_lights), light_intensity [L, 1] or None (the model's scalar)."""
    n_pix = model_input['uv'].shape[1]
    sg = getattr(model, 'render_model', 'sgbasis') == 'sgbasis'
    rgb_all, rough_all, vis_all = [], [], []
    last = None
    chunks = [dict(model_input)] if pixel_chunk is None else split_input(model_input, n_pix, pixel_chunk)
    for l0 in range(0, light_direction.shape[0], light_batch):
        res = []
        for s in chunks:
            s = dict(s)
            s['light_direction'] = light_direction[l0:l0 + light_batch]
            if light_intensity is not None:
                s['light_intensity'] = light_intensity[l0:l0 + light_batch]
            out = model(s, albedo_new=albedo_new, basis_new=basis_new)
            res.append({k: v.detach() for k, v in out.items() if torch.is_tensor(v)})
        last = merge_output(res, n_pix, 1) if len(res) > 1 else {k: (v.reshape(-1) if v.dim() < 3 else v.reshape(-1, v.shape[-1])) for k, v in res[0].items()}
        rgb_all.append(last['sg_rgb_values'].reshape(-1, n_pix, 3))
        rough_all.append(last['sg_specular_rgb_values'].reshape(-1, n_pix, 3))
        if 'visibility' in last:
            vis_all.append(last['visibility'].reshape(-1, n_pix, 3))
    mask = last['network_object_mask'].reshape(n_pix).bool()
    normal = last['normal_pred'] if getattr(model, 'normal_mlp', False) else last['normal_values']
    maps = {'rgb': torch.cat(rgb_all, 0).clamp(0, 1),
            'rough': torch.cat(rough_all, 0) if sg else rough_all[-1][0],
            'mask': mask,
            'normal': normal.reshape(n_pix, 3) * mask[:, None].float(),
            'albedo': last['sg_diffuse_albedo_values'].reshape(n_pix, 3).clamp(0, 1)}
    if vis_all:
        maps['visibility'] = torch.cat(vis_all, 0).clamp(0, 1)
    return maps
