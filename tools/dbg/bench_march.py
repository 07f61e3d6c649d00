"""Ray-march sweep of the stage-1 bench configuration in one process: two-launch dense sweep (psn_sample_points + psn_mlp_infer_pe)
vs the fused sweep (psn_march_sweep) without and with early termination; M = 256 on random pixels (training) and M = 512 on
in-mask pixels (shape_extract).  Prints ms per sweep, evaluated 64-step blocks, and checks the brackets are bit-identical."""
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd.stage1 import NeuralNetwork, Renderer
from psnerf_amd.stage1.rendering import camera_origin, pixel_rays
from psnerf_amd.synthetic import stage1_cfg, stage1_camera
dev = torch.device('cuda:0')
cfg = stage1_cfg('bear')
torch.manual_seed(42)
net = NeuralNetwork(cfg)
ren = Renderer(net, cfg, device=dev)
h, w = 512, 612
K, c2w, S = stage1_camera(cfg, h=h, w=w)
K, c2w = K.to(dev), c2w.to(dev)
g = torch.Generator().manual_seed(0)


def rays_of(pix):
    n = pix.shape[1]
    cam = camera_origin(n, c2w)
    r = pixel_rays(pix.to(dev), K, c2w)
    return cam, r / r.norm(2, 2).unsqueeze(-1)


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, out


res = {}
pix_train = torch.stack([torch.randint(0, w, (4096,), generator=g).float(), torch.randint(0, h, (4096,), generator=g).float()], -1)[None]
# in-mask pixels: a disc of 150 px around the image centre (the object's image has ~179 px radius)
ang, rad = torch.rand(4096, generator=g) * 6.2832, torch.sqrt(torch.rand(4096, generator=g)) * 150
pix_mask = torch.stack([(w / 2 + rad * torch.cos(ang)).round(), (h / 2 + rad * torch.sin(ang)).round()], -1)[None]
for tag, pix, M in (('train_M256_random_pixels', pix_train, 256), ('shape_extract_M512_in_mask_pixels', pix_mask, 512)):
    cam, rays = rays_of(pix)
    args = (cam, rays, 0.5, [M, M + 1], ren.depth_range, cfg['rendering']['radius'], False)
    out = {}
    ref = None
    with torch.no_grad():
        for name, fused, early in (('two_launch_dense', False, False), ('fused_dense', True, False), ('fused_early_exit', True, True)):
            ren.FUSED_SWEEP, ren.EARLY_EXIT = fused, early
            ms, st = timeit(lambda: ren._march_launch(*args))
            if ref is None:
                ref = st
            else:
                assert torch.equal(ref['bracket'], st['bracket']) and torch.equal(ref['flags'], st['flags']), name
            out[name + '_ms'] = round(ms, 3)
        hit = (ref['flags'] & 1).bool()
        out['hit_fraction'] = round(float(hit.float().mean()), 3)
        # opt-in split-bf16 engine (inference_precision = 'bf16x6'): the same three forms, 128-step blocks
        net.inference_precision = 'bf16x6'
        ref6 = None
        for name, fused, early in (('bf16x6_two_launch_dense', False, False), ('bf16x6_fused_dense', True, False), ('bf16x6_fused_early_exit', True, True)):
            ren.FUSED_SWEEP, ren.EARLY_EXIT = fused, early
            ms, st = timeit(lambda: ren._march_launch(*args))
            if ref6 is None:
                ref6 = st
            else:
                assert torch.equal(ref6['bracket'], st['bracket']) and torch.equal(ref6['flags'], st['flags']), name
            out[name + '_ms'] = round(ms, 3)
        out['bf16x6_masks_equal_fp32'] = bool(torch.equal(ref6['flags'], ref['flags']))
        net.inference_precision = 'fp32'
        ren.FUSED_SWEEP = ren.EARLY_EXIT = True
    res[tag] = out
    print(tag, out, flush=True)
print(json.dumps(res))
