"""Occupancy engine (psn_mlp_infer_pe) on 4 M points: the network as it is (softplus, beta = 100) against the SAME pack with
its hidden activations switched to ReLU / none -- isolates what the activation costs the kernel."""
import copy, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from psnerf_amd import hip
from psnerf_amd.stage1 import NeuralNetwork
from psnerf_amd.synthetic import stage1_cfg
dev = torch.device('cuda:0')
cfg = stage1_cfg('bear')
torch.manual_seed(0)
net = NeuralNetwork(cfg).to(dev)
packed = net._occupancy_packed()
Q = 1 << 22
pts = (torch.rand(Q, 3, device=dev) * 2 - 1)
out = torch.empty(Q, 1, device=dev)
def timeit(desc, n=5):
    f = lambda: hip.mlp_infer_pe(desc, packed.w, packed.b, pts, net.octaves_pe, 1.0 / net.rescale, out=out)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
macs = packed.macs_per_row
res = {}
for name, code in (('softplus100', None), ('relu', hip.ACT_RELU), ('none', hip.ACT_NONE)):
    d = copy.deepcopy(packed.desc) if code is not None else packed.desc
    if code is not None:
        for l in range(d.n_layers - 1):
            d.layers[l].act = code
    ms = timeit(d)
    res[name] = (ms, 2.0 * macs * Q / ms / 1e9)
    print('%-12s %7.3f ms  %6.1f TF (true MACs)  %.3f of 157.3' % (name, ms, res[name][1], res[name][1] / 157.3))
