"""March sweep occupancy (1M points): [Q,64] table path (pe_encode + mlp_infer) vs in-kernel encoding (mlp_infer_pe)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
from psnerf_amd.stage1 import NeuralNetwork
from psnerf_amd.synthetic import stage1_cfg
from tests.helpers import stage1_state_dict
dev = torch.device('cuda')
cfg = stage1_cfg('bear') if 'bear' in sys.argv else stage1_cfg('bunny')
net = NeuralNetwork(cfg); net.load_state_dict(stage1_state_dict(cfg, seed=11)); net = net.to(dev)
Q = 1 << 20
p = (torch.rand(Q, 3, device=dev) - 0.5) * 2
packed = net._occupancy_packed()


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    def table():
        tab = hip.pe_encode(p, net.octaves_pe, 64, 1.0 / net.rescale)
        return packed(tab, Q)
    tab0 = hip.pe_encode(p, net.octaves_pe, 64, 1.0 / net.rescale)
    best = {}
    for r in range(3):
        for k, fn in (('table (encode + net)', table), ('net only (table ready)', lambda: packed(tab0, Q)), ('in-kernel encoding', lambda: net.occupancy(p))):
            best[k] = min(best.get(k, 1e9), timeit(fn))
    for k, v in best.items():
        print('%-26s %.3f ms' % (k, v))
    print('equal', torch.equal(table(), net.occupancy(p)))
