"""Stage-1 renderer with the reference's ``Renderer`` interface (stage1/model/rendering.py:9-555,
incl. the phong preview).

Kernel mapping: every occupancy query without a graph (ray-march sweep, secant refinement, shadow-ray light
visibility) goes through the register-resident occupancy engine -- the sweep as one launch over all N x M points, the
first free -> occupied crossing as psn_first_crossing, ALL secant iterations inside one psn_root_find launch, the shadow
rays only for the samples inside the +-1.1 box (psn_shadow_points); the render samples go through ops.GeoFieldFused + the
appearance chains; the transmittance composite is the wave-scan kernel (ops.alpha_composite / hip.composite_fwd); depth
profiles, stratified jitter and sample points are psn_sample_points launches.  Camera rays and the sphere intersection are
a handful of elementwise torch ops on [N, 3] tensors.  The ray march never synchronises with the host; the training
forward does not either when the Trainer selects Renderer._unisurf_sync_free.

All random draws can be injected for parity tests -- ``noise={'miss','hit','nbr'}`` (group-sized tables in the reference's
draw order: the reference-shaped path) or ``noise={'full': [N, S], 'nbr_full': [N, 3]}`` (one row per RAY: the sync-free
training forward, see sync_free_noise_for_reference) --; by default they are drawn on the device.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip, ops

MAX_QUERY_ROWS = 1 << 22  # rows per fused-kernel launch (bounds the positional-encoding table to 1 GiB)


def camera_origin(n_points, world_mat):
    """stage1/model/common.py:205-207."""
    return world_mat[:, :3, -1].unsqueeze(1).repeat(1, n_points, 1)


def pixel_rays(pixels, camera_mat, world_mat):
    """stage1/model/common.py:210-226 -- both pixel axes are divided by fx = K[0,0,0] (reference quirk)."""
    q = (pixels - camera_mat[0, :2, 2]) / camera_mat[0, 0, 0]
    R = world_mat[:, :3, :3]  # d_i = R_i0 qx + R_i1 qy + R_i2 (broadcast products; no library GEMM on the path)
    return q[..., 0:1] * R[:, None, :, 0] + q[..., 1:2] * R[:, None, :, 1] + R[:, None, :, 2]


def sphere_intersection(cam_loc, ray_dirs, r=1.0):
    """stage1/model/rendering.py:576-596."""
    n_img, n_pix, _ = ray_dirs.shape
    b = (ray_dirs * cam_loc.unsqueeze(1)).sum(-1).reshape(-1)
    under = b ** 2 - (cam_loc.norm(2, 1).reshape(-1, 1).expand(n_img, n_pix).reshape(-1) ** 2 - r ** 2)
    hit = under > 0
    root = torch.sqrt(under.clamp(min=0))
    out = torch.stack([-root - b, root - b], dim=-1)
    out = torch.where(hit.unsqueeze(-1), out, torch.zeros_like(out))
    return out.reshape(n_img, n_pix, 2).clamp_min(0.0), hit.reshape(n_img, n_pix)


def finite_mask(t):
    return (t.abs() != np.inf) & ~torch.isnan(t)


def is_per_ray_noise(noise):
    """True when the injected draws (if any) are the per-ray tables of the sync-free forward."""
    return not noise or all(k in ('full', 'nbr_full') for k in noise)


def sync_free_noise_for_reference(noise, hit_mask):
    """Per-ray tables {'full': [N, S], 'nbr_full': [N, 3]} -> the group-sized tables of the reference's draw order
    (rendering.py:139 miss rays, :163 hit rays, :204 neighbour offsets of the hit points), i.e. the draws that make the
    reference-shaped path / the oracle use the same number for the same (ray, sample) as the sync-free forward."""
    hit = hit_mask.reshape(-1).bool()
    out = {}
    if noise.get('full') is not None:
        full = noise['full'].reshape(hit.shape[0], -1)
        out['miss'] = full[~hit].unsqueeze(0)
        out['hit'] = full[hit].unsqueeze(0)
    if noise.get('nbr_full') is not None:
        out['nbr'] = noise['nbr_full'][hit]
    return out


class Renderer(nn.Module):
    def __init__(self, model, cfg_all, device=None, **kwargs):
        super().__init__()
        cfg = cfg_all['rendering']
        self._device = device
        self.depth_range = [cfg['near'], cfg['far']]
        self.n_max_network_queries = cfg['n_max_network_queries']
        self.white_background = cfg['white_background']
        self.cfg = cfg
        self.model = model.to(device) if device is not None else model

    def forward(self, pixels, camera_mat, world_mat, scale_mat, rendering_technique, add_noise=True, eval_=False,
                it=0, visibility=False, light_dir=None, noise=None):
        if rendering_technique == 'unisurf':
            return self.unisurf(pixels, camera_mat, world_mat, scale_mat, it=it, add_noise=add_noise, eval_=eval_,
                                noise=noise)
        if rendering_technique == 'shape_extract':
            return self.shape_extract(pixels, camera_mat, world_mat, scale_mat, it=it, visibility=visibility,
                                      light_dir=light_dir)
        if rendering_technique == 'phong_renderer':
            return self.phong_renderer(pixels, camera_mat, world_mat, scale_mat)
        raise ValueError('Choose unisurf or shape_extract')

    def _u(self, n, dev):
        """(linspace(0, 1, n), 1 - linspace(0, 1, n)) on the device, cached: the tables of csrc/sample.hip."""
        cache = self.__dict__.setdefault('_u_cache', {})
        key = (int(n), str(dev))
        if key not in cache:
            u = torch.linspace(0.0, 1.0, steps=int(n), device=dev)
            cache[key] = (u.contiguous(), (1.0 - u).contiguous())
        return cache[key]

    # ---- occupancy queries without a graph -------------------------------------------------------
    def _occ(self, pts):
        """sigmoid(-10 logit) for [Q,3] points, fused kernel, chunked."""
        outs = []
        for s in range(0, pts.shape[0], MAX_QUERY_ROWS):
            outs.append(self.model.occupancy(pts[s:s + MAX_QUERY_ROWS]))
        return torch.cat(outs, 0) if len(outs) != 1 else outs[0]

    # ---- stage1/model/rendering.py:410-523 -------------------------------------------------------
    @torch.no_grad()
    def ray_marching(self, ray0, ray_direction, model=None, c=None, tau=0.5, n_steps=(128, 129), n_secant_steps=8,
                     depth_range=(25, 40), max_points=3500000, rad=1.0, clip=False):
        state = self._march_launch(ray0, ray_direction, tau, n_steps, depth_range, rad, clip)
        return self._finish(state, n_secant_steps)

    COMPACT_SECANT = False  # True: the round-1 secant (host-synchronised compaction + per-iteration launches)
    FUSED_SWEEP = True      # False: the two-launch sweep (psn_sample_points + psn_mlp_infer_pe on an [N, M, 3] point tensor)
    EARLY_EXIT = True       # False: every block of every ray is evaluated (the dense sweep; bit-identity tests)

    def _finish(self, state, n_secant_steps):
        return self._march_finish_compact(state, n_secant_steps) if self.COMPACT_SECANT else self._march_finish(state, n_secant_steps)

    @torch.no_grad()
    def _march_launch(self, ray0, ray_direction, tau, n_steps, depth_range, rad, clip, far=None):
        """First half of ray_marching (rendering.py:410-480): everything up to the first-crossing mask -- all of it
        asynchronous device work (the 256-step occupancy sweep is one fused launch), no host synchronisation.  ``far``:
        the sphere exit depths when the caller has them already (psn_stage1_rays)."""
        B, N, _ = ray0.shape
        dev = ray0.device
        n_steps = int(n_steps[0])  # the reference draws randint(n, n+1): a constant
        if far is None:
            far = sphere_intersection(ray0[:, 0], ray_direction, r=rad)[0][..., 1].contiguous()
        u = self._u(n_steps, dev)
        m = self.model
        x3 = getattr(m, 'inference_precision', 'fp32') == 'bf16x6'
        if (self.FUSED_SWEEP and x3 and not clip and n_steps % 128 == 0 and ray0.is_cuda and hasattr(m, '_occupancy_packed_x3')
                and m._hidden_is_256() and not torch.is_grad_enabled()):
            # opt-in split-bf16 engine: the same one-launch sweep (points formed in the kernel, early exit per 128-step block)
            occ, _ = m._occupancy_packed_x3().march_sweep(ray0.reshape(-1, 3).contiguous(), ray_direction.reshape(-1, 3).contiguous(),
                                                          far.reshape(-1), u[0], u[1], float(depth_range[0]), n_steps, tau, m.octaves_pe,
                                                          1.0 / m.rescale, early_exit=self.EARLY_EXIT)
        elif (self.FUSED_SWEEP and not clip and n_steps % 64 == 0 and ray0.is_cuda and hasattr(m, '_occupancy_packed')
                and m._hidden_is_256() and not x3):
            # (with the opt-in split-bf16 engine and a step count that is no multiple of 128 the sweep takes the two-launch form below)
            # one launch: sweep points generated and encoded in the occupancy kernel, a workgroup = 64 consecutive steps of
            # one ray, and the blocks behind a ray's first sign change are not evaluated (psn_march_sweep; the reference's
            # result depends on nothing behind it, rendering.py:472-504)
            packed = m._occupancy_packed(allow_x3=True)  # ('bf16x3': the same kernel on split-bf16 weight stages; otherwise the exact pack)
            ops._hit('march_sweep')
            occ, _ = hip.march_sweep(packed.desc, packed.w, packed.b, ray0.reshape(-1, 3).contiguous(),
                                     ray_direction.reshape(-1, 3).contiguous(), far.reshape(-1), u[0], u[1], float(depth_range[0]),
                                     n_steps, tau, m.octaves_pe, 1.0 / m.rescale, early_exit=self.EARLY_EXIT,
                                     macs_per_row=getattr(packed, 'macs_per_row', None))
        else:
            if self.FUSED_SWEEP:  # (False: the two-launch cross-check of the tests)
                ops.fallback('stage1 ray-march sweep -> point table + occupancy launch', ray0,
                             'n_steps %d (64 | n_steps needed), clip %s, hidden width' % (n_steps, bool(clip)))
            # sweep points ray0 + dir * (near (1 - t) + far t), t = linspace(0, 1, n_steps): one launch (csrc/sample.hip)
            p_prop = torch.empty(B * N, n_steps, 3, device=dev)
            hip.sample_points(ray0.reshape(-1, 3).contiguous(), ray_direction.reshape(-1, 3).contiguous(),
                              far.reshape(-1), p_prop, False, float(depth_range[0]), u)
            occ = self._occ(p_prop.reshape(-1, 3)).view(B * N, n_steps)
        if clip:
            pp4 = p_prop.view(B * N, n_steps, 3)
            occ = occ.clone()
            occ[(pp4 > 1).any(-1)] = tau - 1
            occ[(pp4 < -1).any(-1)] = tau - 1
        # first free -> occupied crossing and its bracket (rendering.py:457-504): one launch, one wave per ray
        bracket, flags = hip.first_crossing(occ.contiguous(), far.reshape(-1).contiguous(), u[0], u[1],
                                            float(depth_range[0]), tau)
        return dict(ray0=ray0, ray_direction=ray_direction, tau=tau, far=far, bracket=bracket, flags=flags)

    @torch.no_grad()
    def _march_finish(self, st, n_secant_steps):
        """Second half (rendering.py:480-523): the secant refinement of EVERY ray in one launch (rays without a crossing
        carry a benign bracket and are overwritten below), no compaction and therefore no host synchronisation: the
        refinement is latency-bound -- one serial pass through the network per iteration, whatever the number of rays
        up to 64 per CU -- so masked rays cost nothing, while nonzero() cost a round trip to the host."""
        ray0, ray_direction = st['ray0'], st['ray_direction']
        B, N, _ = ray0.shape
        mask = (st['flags'] & 1).bool()
        first_free = (st['flags'] & 2).bool()
        d_pred = self._march_root(st, n_secant_steps)
        out = torch.where(mask, d_pred, torch.full_like(d_pred, float('inf')))
        out = torch.where(first_free, out, torch.zeros_like(out))
        return out.view(B, N)

    @torch.no_grad()
    def _march_root(self, st, n_secant_steps):
        """Refined depth of every ray's bracket (meaningful where flags & 1)."""
        return self._root_find(st['bracket'], st['ray0'].reshape(-1, 3).contiguous(), st['ray_direction'].reshape(-1, 3).contiguous(),
                               st['tau'], n_secant_steps)

    @torch.no_grad()
    def _march_finish_compact(self, st, n_secant_steps):
        """The round-1 formulation of the second half, kept as a cross-check of the fused root finder (tests) and for
        A/B timing: nonzero() compacts the rays with a crossing (a host synchronisation), then one encoding + network +
        update launch per secant iteration on the compacted rays."""
        ray0, ray_direction = st['ray0'], st['ray_direction']
        B, N, _ = ray0.shape
        dev = ray0.device
        mask = (st['flags'] & 1).bool()
        first_free = (st['flags'] & 2).bool()
        mi = mask.nonzero(as_tuple=True)[0]
        b = st['bracket'][:, mi]
        d_low, d_high, f_low, f_high = (b[i].contiguous().clone() for i in range(4))
        origin, direction = ray0.reshape(-1, 3)[mi].contiguous(), ray_direction.reshape(-1, 3)[mi].contiguous()
        d_pred = torch.empty_like(d_low)
        if mi.numel() > 0:
            p_mid = torch.empty(d_low.shape[0], 3, device=dev)
            hip.secant_step(None, st['tau'], d_pred, d_low, d_high, f_low, f_high, origin, direction, p_mid)
            for i in range(n_secant_steps):
                occ = self._occ(p_mid)[..., 0].contiguous()
                hip.secant_step(occ, st['tau'], d_pred, d_low, d_high, f_low, f_high, origin, direction,
                                p_mid if i + 1 < n_secant_steps else None)
        out = torch.full((B * N,), float('inf'), device=dev)
        out[mi] = d_pred
        return torch.where(first_free, out, torch.zeros_like(out)).view(B, N)

    def _root_find(self, bracket, origin, direction, tau, n_iter):
        m = self.model
        packed = m._occupancy_packed()
        ops._hit('root_find')
        return hip.root_find(packed.desc, packed.w, packed.b, origin, direction, bracket, tau, n_iter, m.octaves_pe,
                             1.0 / m.rescale)

    # ---- stage1/model/rendering.py:525-555 -------------------------------------------------------
    @torch.no_grad()
    def secant(self, f_low, f_high, d_low, d_high, n_secant_steps, ray0_masked, ray_direction_masked, tau, it=0):
        if f_low.is_cuda and f_low.numel() > 0 and hasattr(self.model, '_occupancy_packed') and self.model._hidden_is_256():
            # all iterations in one launch (psn_root_find)
            bracket = torch.stack([d_low, d_high, f_low, f_high]).float().contiguous()
            return self._root_find(bracket, ray0_masked.contiguous(), ray_direction_masked.contiguous(), tau, n_secant_steps)
        if f_low.numel() > 0:
            ops.fallback('stage1 secant -> one launch set per iteration', f_low, 'hidden width is not 256')
        d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low
        if d_pred.numel() == 0:
            return d_pred
        for _ in range(n_secant_steps):
            p_mid = ray0_masked + d_pred.unsqueeze(-1) * ray_direction_masked
            f_mid = self._occ(p_mid)[..., 0] - tau
            lo = f_mid < 0
            d_low = torch.where(lo, d_pred, d_low)
            f_low = torch.where(lo, f_mid, f_low)
            d_high = torch.where(lo, d_high, d_pred)
            f_high = torch.where(lo, f_high, f_mid)
            d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low
        return d_pred

    FUSED_GLUE = True  # False: the torch formulation of the per-ray set-up / surface points (tests compare the two)

    def _surface_launch(self, pixels, camera_mat, world_mat, ray_steps):
        B, N, _ = pixels.shape
        far = None
        if (self.FUSED_GLUE and pixels.is_cuda and B == 1 and tuple(world_mat.shape[1:]) == (4, 4) and pixels.dtype == torch.float32
                and pixels.is_contiguous()):
            # origins, normalised directions and sphere exit depths in one launch (psn_stage1_rays) instead of ~25
            cam, rays, far = hip.stage1_rays(pixels[0], camera_mat[0].float().contiguous(), world_mat[0].float().contiguous(), self.cfg['radius'])
            cam, rays, far = cam.unsqueeze(0), rays.unsqueeze(0), far.unsqueeze(0)
        else:
            cam = camera_origin(N, world_mat)
            rays = pixel_rays(pixels, camera_mat, world_mat)
            rays = rays / rays.norm(2, 2).unsqueeze(-1)
        state = self._march_launch(cam, rays, 0.5, [int(ray_steps), int(ray_steps) + 1], self.depth_range,
                                   self.cfg['radius'], False, far=far)
        return cam, rays, state

    @torch.no_grad()
    def prefetch_surface(self, pixels, camera_mat, world_mat):
        """Queue the ray-march sweep for ``pixels`` now; the next forward() with the SAME pixel tensor picks it up.  The
        trainer calls this before its data-only host work (ground-truth gathers, mask counts), which then overlaps
        with the 10 ms sweep instead of preceding it."""
        steps = int(self.cfg['ray_marching_steps'])
        self._pref = (pixels, steps, self._surface_launch(pixels, camera_mat, world_mat, steps))
        if hasattr(self.model, 'prepack'):
            self.model.prepack(chains=True)  # the host has ~10 ms to spare while the sweep runs

    def _surface(self, pixels, camera_mat, world_mat, ray_steps):
        """Shared prologue of unisurf / shape_extract (rendering.py:67-108, 311-340)."""
        B, N, _ = pixels.shape
        pref, self._pref = getattr(self, '_pref', None), None
        if pref is not None and pref[0] is pixels and pref[1] == int(ray_steps):
            cam, rays, state = pref[2]  # requested earlier by prefetch_surface: the sweep is already running
        else:
            cam, rays, state = self._surface_launch(pixels, camera_mat, world_mat, ray_steps)
        self._last_far = state['far'].reshape(-1)  # sphere exit depth per ray (rendering.py:576-596), reused by unisurf
        if self.FUSED_GLUE and pixels.is_cuda and B == 1 and not self.COMPACT_SECANT:
            # d_i, its masks and the surface points in one launch (psn_surface_points) instead of ~18
            cam, rays = cam.reshape(-1, 3), rays.reshape(-1, 3)
            with torch.no_grad():
                dists, obj_mask, points = hip.surface_points(self._march_root(state, 8), state['flags'], cam, rays)
            return cam, rays, dists, obj_mask, points
        d_i = self._finish(state, 8)
        zero_occ = d_i == 0
        ok = finite_mask(d_i)
        dists = torch.where(ok, d_i, torch.ones_like(d_i))
        dists = torch.where(zero_occ, torch.zeros_like(dists), dists)
        obj_mask = (ok & ~zero_occ)[0]
        dists = dists[0]
        cam = cam.reshape(-1, 3)
        rays = rays.reshape(-1, 3)
        points = (cam + rays * dists.unsqueeze(-1)).view(-1, 3)
        return cam, rays, dists, obj_mask, points

    # ---- stage1/model/rendering.py:50-226 --------------------------------------------------------
    def unisurf(self, pixels, camera_mat, world_mat, scale_mat, add_noise=False, it=100000, eval_=False, noise=None):
        noise = noise or {}
        B, N, _ = pixels.shape
        dev = pixels.device
        cfg = self.cfg
        steps, steps_out = cfg['num_points_in'], cfg['num_points_out']
        near = float(self.depth_range[0])

        cam, rays, dists, obj_mask, points = self._surface(pixels, camera_mat, world_mat, cfg['ray_marching_steps'])
        far = self._last_far  # = sphere_intersection(cam, rays, r)[..., 1], already computed for the sweep
        if self.sync_free and not eval_ and is_per_ray_noise(noise) and near > 0 and pixels.is_cuda:
            return self._unisurf_sync_free(cam, rays, dists, obj_mask, points, far, it, add_noise, noise)

        # hit / miss ray lists ONCE (two nonzero() = the data-dependent host synchronisations of this function); every
        # gather / scatter below is an index op with them instead of a boolean mask (each of which would sync again)
        hit_idx = obj_mask.nonzero(as_tuple=True)[0]
        miss_idx = (~obj_mask).nonzero(as_tuple=True)[0]
        if 'full' in noise or 'nbr_full' in noise:  # per-ray tables (the sync-free forward's form) -> group-sized tables
            noise = dict(noise, **sync_free_noise_for_reference(noise, obj_mask))
        delta = float(torch.max(cfg['interval_start'] * torch.exp(-1 * cfg['interval_decay'] * it * torch.ones(1)),
                                cfg['interval_end'] * torch.ones(1)))  # fp32 like rendering.py:116-117
        if near > 0:
            nonzero_dnp = True  # dnp = max(d - delta, near) >= near > 0: the reference's (dnp != 0).all() holds
        else:
            dnp_t = dists[hit_idx] - delta
            nonzero_dnp = bool((torch.where(dnp_t < near, torch.full_like(dnp_t, near), dnp_t) != 0.0).all())
        full_steps = steps + steps_out if (nonzero_dnp and it > 5000) else steps

        def draw(key, n_rays):  # stratified-jitter noise in the reference's draw order (miss rays, then hit rays)
            if not add_noise:
                return None
            nz = noise.get(key)
            if nz is None:
                nz = torch.rand(B, n_rays, full_steps, device=dev)
            return nz.to(dev).reshape(-1).contiguous()

        # depth profiles, jitter and points of both ray groups: two launches of csrc/sample.hip (rendering.py:110-176)
        p_fg = torch.empty(B * N, full_steps, 3, device=dev)
        hip.sample_points(cam, rays, far, p_fg, False, near, self._u(full_steps, dev), idx=miss_idx,
                          noise=draw('miss', miss_idx.shape[0]))
        if full_steps != steps:
            hip.sample_points(cam, rays, far, p_fg, True, near, self._u(steps_out, dev), idx=hit_idx, dist=dists, delta=delta,
                              u1=self._u(steps, dev), noise=draw('hit', hit_idx.shape[0]))
        else:
            hip.sample_points(cam, rays, far, p_fg, True, near, self._u(steps, dev), idx=hit_idx, dist=dists, delta=delta,
                              noise=draw('hit', hit_idx.shape[0]))
        p_fg = p_fg.reshape(-1, 3)
        view = (-1 * rays).unsqueeze(-2).expand(-1, full_steps, -1).reshape(-1, 3)

        # (same RNG order as the reference: the model call below draws nothing)
        surf = points[hit_idx]
        n_surf = surf.shape[0]
        if not eval_:
            nz = noise.get('nbr')
            if nz is None:
                nz = torch.rand_like(surf)
            pp = torch.cat([surf, surf + (nz.to(dev) - 0.5) * 0.01], dim=0)
        else:
            pp = surf

        rgb, alpha = self.model(p_fg, view, return_addocc=True)  # one launch chain; no 64000-point chunking needed
        rgb = rgb.reshape(B * N, full_steps, 3)
        alpha = alpha.reshape(B * N, full_steps)
        rgb_values, acc = ops.alpha_composite(alpha, rgb, bool(self.white_background))
        norm_pred = torch.zeros(B * N, 3, device=dev)
        diff_norm = None
        if n_surf > 0:
            g = self.model.gradient(pp)[:, 0, :]
            nrm = g / (g.norm(2, dim=1).unsqueeze(-1) + 10 ** (-5))
            norm_pred[hit_idx] = nrm[:n_surf]
            if not eval_:
                diff_norm = torch.norm(nrm[:n_surf] - nrm[n_surf:], dim=-1)
        elif not eval_:
            diff_norm = torch.zeros(0, device=dev)
        return {
            'rgb': rgb_values.reshape(B, -1, 3),
            'mask_pred': obj_mask,
            'diff_norm': diff_norm,
            'normal_pred': norm_pred.reshape(B, -1, 3),
            'acc_map': acc.reshape(B, -1),
        }

    sync_free = False  # set by the Trainer: training forward without host synchronisation (see _unisurf_sync_free)

    def _unisurf_sync_free(self, cam, rays, dists, obj_mask, points, far, it, add_noise, noise=None):
        """The training forward of unisurf() (rendering.py:110-224) without a single host synchronisation.  The
        reference-shaped path above needs the hit / miss ray LISTS (two nonzero() calls) because it samples the two
        groups separately, evaluates the surface normals on the compacted hit points and returns a compact
        ``diff_norm [N_hit]``.  Here both groups are sampled by one launch that reads the per-ray hit flag
        (psn_sample_points_flagged), the normals are evaluated for ALL rays (2 N instead of 2 N_hit points next to N S
        render samples: +1 %) and masked, and the smoothness term is returned as ``diff_norm_full [N]`` + ``mask_pred``
        for a masked sum over a device-resident count (Loss).  Same arithmetic per ray; the random draws have the
        reference's distribution but not its stream order (group-sized draws need the group sizes on the host).
        ``noise={'full': [N, S] stratified-jitter table, 'nbr_full': [N, 3] neighbour offsets}`` injects the draws, one row
        per ray; ``sync_free_noise_for_reference`` turns such tables into the group-sized tables ('miss', 'hit', 'nbr') that
        make the reference / the oracle draw the same numbers for the same rays."""
        noise = noise or {}
        cfg = self.cfg
        dev = cam.device
        N = cam.shape[0]
        steps, steps_out = cfg['num_points_in'], cfg['num_points_out']
        near = float(self.depth_range[0])
        delta = float(torch.max(cfg['interval_start'] * torch.exp(-1 * cfg['interval_decay'] * it * torch.ones(1)),
                                cfg['interval_end'] * torch.ones(1)))
        full_steps = steps + steps_out if it > 5000 else steps  # near > 0: (dnp != 0).all() holds (rendering.py:124)
        nz = None
        if add_noise:
            nz = noise.get('full')
            if nz is None:
                nz = torch.rand(N * full_steps, device=dev)
            else:
                if nz.numel() != N * full_steps:
                    raise ValueError("noise['full'] must hold N x S = %d x %d values, got %s" % (N, full_steps, tuple(nz.shape)))
                nz = nz.to(dev, torch.float32).reshape(-1).contiguous()
        p_fg = torch.empty(N, full_steps, 3, device=dev)
        flags = obj_mask.contiguous()
        if full_steps != steps:
            hip.sample_points_flagged(cam, rays, dists.contiguous(), far, flags, p_fg, near, delta, self._u(steps_out, dev),
                                      self._u(steps, dev), self._u(full_steps, dev), noise=nz)
        else:
            hip.sample_points_flagged(cam, rays, dists.contiguous(), far, flags, p_fg, near, delta, self._u(steps, dev), None,
                                      self._u(full_steps, dev), noise=nz)
        p_fg = p_fg.reshape(-1, 3)
        view = (-1 * rays).unsqueeze(-2).expand(-1, full_steps, -1).reshape(-1, 3)
        nbr = noise.get('nbr_full')
        if nbr is None:
            nbr = torch.rand_like(points)
        elif tuple(nbr.shape) != tuple(points.shape):
            raise ValueError("noise['nbr_full'] must be [N, 3] = %s, got %s" % (tuple(points.shape), tuple(nbr.shape)))
        pp = torch.cat([points, points + (nbr.to(dev) - 0.5) * 0.01], dim=0)  # every ray; masked below
        if hasattr(self.model, 'render_and_gradient'):
            # the 2 N normal points ride behind the N S render samples through the geometry network (one set of launches)
            rgb, alpha, g = self.model.render_and_gradient(p_fg, view, pp)
            g = g[:, 0, :]
        else:
            rgb, alpha = self.model(p_fg, view, return_addocc=True)
            g = self.model.gradient(pp)[:, 0, :]
        rgb_values, acc = ops.alpha_composite(alpha.reshape(N, full_steps), rgb.reshape(N, full_steps, 3),
                                              bool(self.white_background))
        if self.FUSED_GLUE and g.is_cuda:
            norm_pred, diff_full = ops.SurfaceNormals.apply(g, flags)  # one launch each way instead of ~10 / ~25
        else:
            nrm = g / (g.norm(2, dim=1).unsqueeze(-1) + 10 ** (-5))
            norm_pred = torch.where(flags.unsqueeze(-1), nrm[:N], torch.zeros_like(nrm[:N]))
            diff_full = torch.norm(nrm[:N] - nrm[N:], dim=-1)
        return {'rgb': rgb_values.reshape(1, -1, 3), 'mask_pred': obj_mask, 'diff_norm': None, 'diff_norm_full': diff_full,
                'normal_pred': norm_pred.reshape(1, -1, 3), 'acc_map': acc.reshape(1, -1)}

    # ---- stage1/model/rendering.py:297-376 -------------------------------------------------------
    @torch.no_grad()
    def shape_extract(self, pixels, camera_mat, world_mat, scale_mat, it=100000, visibility=False, light_dir=None):
        B, N, _ = pixels.shape
        dev = pixels.device
        self.model.eval()
        cam, rays, dists, obj_mask, points = self._surface(pixels, camera_mat, world_mat, 512)
        surf = points[obj_mask]
        normal = torch.zeros(B * N, 3, device=dev)
        if len(surf) > 0:
            g = torch.cat([self.model.gradient(ps, tflag=False)[:, 0, :] for ps in torch.split(surf, 1000000, dim=0)], 0)
            normal[obj_mask] = F.normalize(g, dim=-1)
        out = {'mask': obj_mask.reshape(B, -1), 'normal': normal.reshape(B, -1, 3), 'points': points.reshape(B, -1, 3)}
        if visibility and light_dir is not None:
            vis = torch.ones(light_dir.shape[0], N, device=dev)
            if len(surf) > 0:
                chunks = [self.light_visibility(surf=surf, light_dir=light_dir[s:s + 96])
                          for s in range(0, len(light_dir), 96)]
                vis[obj_mask[None].expand_as(vis)] = torch.cat(chunks, dim=0)
            out['visibility'] = vis
        return out

    # ---- stage1/model/rendering.py:228-293 -------------------------------------------------------
    @torch.no_grad()
    def phong_renderer(self, pixels, camera_mat, world_mat, scale_mat):
        """The shaded preview of the current shape (rendering.py:228-293; what training.py:62-118 renders for its image grids):
        the 512-step march + root finder of shape_extract, unit normals grad / |grad| at the surface points (no epsilon, :282), one
        light at the camera (:240-241), rgb = min(0.3 + 0.7 max(n . l, 0), 1) on the surface and 1 elsewhere."""
        B, N, _ = pixels.shape
        dev = pixels.device
        self.model.eval()
        cam, rays, dists, obj_mask, points = self._surface(pixels, camera_mat, world_mat, 512)
        cam_o = cam.reshape(-1, 3)[0]
        light = (cam_o / cam_o.norm(2)).unsqueeze(1)
        rgb = torch.ones(B * N, 3, device=dev)
        surf = points[obj_mask]
        if len(surf) > 0:
            g = torch.cat([self.model.gradient(ps, tflag=False)[:, 0, :] for ps in torch.split(surf, 1000000, dim=0)], 0)
            n = g / g.norm(2, 1, keepdim=True)
            diffuse = (n * light[:, 0]).sum(-1, keepdim=True).clamp_min(0).repeat(1, 3) * 0.7   # (a [Ns,3] x [3,1] product without a library GEMM)
            rgb[obj_mask] = (0.3 + diffuse).clamp_max(1.0)
        return {'rgb': rgb.reshape(B, -1, 3)}

    # ---- stage1/model/rendering.py:378-408 -------------------------------------------------------
    @torch.no_grad()
    def light_visibility(self, surf=None, light_dir=None, lnear=0.1, lfar=3.5, tau=0.5, n_steps=128,
                         max_points=3500000):
        """Shadow-ray transmittance from every surface point toward every light: 1 - acc, [L*Ns]."""
        dev = surf.device
        L, Ns = light_dir.shape[0], surf.shape[0]
        surf = surf.contiguous()
        u = self._u(n_steps, dev)
        per_light = Ns * n_steps
        # Dense rows per launch group: the 12-byte points and 8-byte row numbers of the worst case (every sample inside
        # the box); the 256-byte encodings are only built for the rows that survive the box test.
        lights_per_chunk = max(1, (4 * MAX_QUERY_ROWS) // max(per_light, 1))
        outs = []
        for l0 in range(0, L, lights_per_chunk):
            ld = light_dir[l0:l0 + lights_per_chunk].contiguous()
            n_rays = ld.shape[0] * Ns
            # points of the samples that lie inside the +-1.1 box, compacted (csrc/sample.hip): the others have
            # occupancy 0 by rendering.py:400-401 and never reach the network
            pts, rows, counter = hip.shadow_points(surf, ld, n_steps, lnear, lfar, u[0], u[1], 1.1)
            alpha = torch.zeros(n_rays * n_steps, device=dev)
            m = self.model
            if self.SHADOW_SYNC_FREE and hasattr(m, '_occupancy_packed') and m._hidden_is_256():
                # the network runs over the compacted list straight away: its length stays on the device (workgroups behind it
                # leave at once) and the occupancies are scattered to their (ray, step) slots by the kernel itself
                ops._hit('shadow_indirect')
                m._occupancy_packed(allow_x3=True).on_points(pts, m.octaves_pe, 1.0 / m.rescale, out=alpha, n_rows_dev=counter, out_rows=rows)
            else:
                if self.SHADOW_SYNC_FREE:
                    ops.fallback('stage1 shadow rays -> host-synchronised compaction', surf, 'hidden width is not 256')
                n_in = int(counter.item())  # one host synchronisation per launch group
                if n_in > 0:
                    alpha.index_copy_(0, rows[:n_in], self._occ(pts[:n_in]).reshape(-1))
            _, _, acc = hip.composite_fwd(alpha.view(n_rays, n_steps), None, False, need_weights=False)
            outs.append(1 - acc)
            self._shadow_stats = (counter, n_rays * n_steps)
        return torch.cat(outs, 0)

    SHADOW_SYNC_FREE = True  # False: host-synchronised launch sizes (one counter.item() per launch group) + index_copy_

    @property
    def last_shadow_stats(self):
        """(in-box samples, all samples) of the last launch group of light_visibility (reads the device counter: synchronises)."""
        counter, n_all = self._shadow_stats
        return int(counter.item()), n_all
