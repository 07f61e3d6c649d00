"""psn_mlp_infer vs psn_mlp_infer_padded on the stage-2 visibility rows of a 4096-px rank shard (L = 96, V = 8, 3686 real rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd import ops, fused
dev = torch.device('cuda:0')
ns, live, L, V = 4096, 3686, 96, 8
g = torch.Generator().manual_seed(0)
pe_x = torch.randn(ns, 64, generator=g).to(dev)
pe_l = torch.randn(L + V, 64, generator=g).to(dev)
dims = [(256, 78)] + [(256, 256)] * 2 + [(256, 256 + 78)] + [(256, 256)] + [(1, 256)]
params = []
for o, i in dims:
    params += [(torch.randn(o, i, generator=g) / i ** 0.5).to(dev), (torch.randn(o, generator=g) * 0.1).to(dev)]
c = torch.arange(39)
cols = torch.cat([c, 64 + c]).to(dev)
packed = fused.pack_relu_mlp(params[0::2], params[1::2], 39, 39, 2)
cnt = torch.tensor([float(live)], device=dev)
full = torch.tensor([float(ns)], device=dev)
half = torch.tensor([float(ns // 2)], device=dev)
for name, kw in (('plain', {}), ('padded', dict(live_count=cnt)), ('map-only', dict(live_count=full)), ('half', dict(live_count=half)), ('plain', {}), ('padded', dict(live_count=cnt)), ('map-only', dict(live_count=full)), ('exact', None)):
    x = pe_x if kw is not None else pe_x[:live].contiguous()
    kw = kw or {}
    for _ in range(3): ops.VisibilityPair.launch(x, pe_l, L, cols, 2, params, True, packed=packed, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.VisibilityPair.launch(x, pe_l, L, cols, 2, params, True, packed=packed, **kw)
    e1.record(); torch.cuda.synchronize()
    print('%-7s %.3f ms / launch group' % (name, e0.elapsed_time(e1) / 20))
