import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, faulthandler
faulthandler.enable()
from tests.test_graph_gpu import _make
from psnerf_amd.synthetic import stage2_inputs
from psnerf_amd.stage2.graph import GraphedTrainStep
cuda = torch.device('cuda:0')
N, L, V, n_it = 2560, 10, 4, 9
step, NL = _make(cuda, 5001)
run = GraphedTrainStep(step, warmup=1, pad_multiple=512)
for it in range(n_it):
    inp, gt = stage2_inputs(N, L, V, seed=500 + it, surface_frac=(0.55, 0.57, 0.75, 0.56, 0.76, 0.58, 0.74, 0.55, 0.77)[it])
    inp['surface_idx'] = inp['surface_mask'][0].nonzero(as_tuple=True)[0]
    ns = int(inp['surface_idx'].numel()); cap = -(-ns // 512) * 512
    l_slt = torch.randperm(NL, generator=torch.Generator().manual_seed(it))[:L].to(cuda)
    nz = torch.zeros(cap, 3); nz[:ns] = torch.randn(ns, 3, generator=torch.Generator().manual_seed(50 + it)) * 0.01
    print('it', it, 'ns', ns, 'cap', cap, flush=True)
    terms, _ = run.step({k: v.to(cuda) for k, v in inp.items()}, {k: v.to(cuda) for k, v in gt.items()}, l_slt, train_order=False, noise={'xyz': nz.to(cuda)})
    torch.cuda.synchronize()
    print('   loss', float(terms['total'].detach()), run.n_eager, run.n_captures, run.n_replays, flush=True)
