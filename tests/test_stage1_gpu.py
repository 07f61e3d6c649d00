"""Stage-1 product path (HIP) against golden vectors captured from the reference and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, assert_close, grad_digest, stage1_state_dict, state_dict_digest, stage1_cfg

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize('tag,over', [('h64', {'model.hidden_dim': 64, 'model.feat_size': 64}), ('h256', {})])
def test_network_golden(cuda, tag, over):
    from psnerf_amd.stage1 import NeuralNetwork
    g = np.load(os.path.join(GOLDEN, 'stage1_net_%s.npz' % tag))
    cfg = stage1_cfg('bunny', **over)
    sd = stage1_state_dict(cfg, seed=11)
    assert state_dict_digest(sd) == str(g['sd_digest'])
    net = NeuralNetwork(cfg)
    net.load_state_dict(sd)
    net.to(cuda)
    p, ray_d = T(g['p'], cuda), T(g['ray_d'], cuda)
    occ = net.infer_occ(p.clone())
    grad = net.gradient(p.clone())[:, 0]
    rgb, alpha = net(p.clone(), ray_d, return_addocc=True)
    assert_close(occ.detach().cpu(), g['occ'], 1e-4, 'occ')
    assert_close(grad.detach().cpu(), g['grad'], 1e-4, 'grad')
    assert_close(rgb.detach().cpu(), g['rgb'], 1e-4, 'rgb')
    assert_close(alpha.detach().cpu(), g['alpha'], 1e-4, 'alpha')
    with torch.no_grad():
        assert_close(net(p, only_occupancy=True).cpu(), g['occ_only'], 1e-4, 'occ_only (fused or GEMM path)')
        assert_close(net(p, return_logits=True).cpu(), g['logits'], 1e-4, 'logits')
    loss = (rgb * T(g['c_rgb'], cuda)).sum() + (alpha * T(g['c_alpha'], cuda)).sum() \
        + (occ * T(g['c_occ'], cuda)).sum() * 0.01 + (grad * T(g['c_grad'], cuda)).sum() * 0.1
    loss.backward()
    assert_close(float(loss.detach()), float(g['loss']), 1e-4, 'loss')
    names, norms, projs = grad_digest({k: v.grad for k, v in net.named_parameters()})
    assert names == list(g['grad_names'])
    if tag == 'h64':
        for k, v in net.named_parameters():
            assert_close(v.grad.cpu(), g['g_' + k], 1e-3, 'grad ' + k)
    assert_close(norms, g['grad_norms'], 1e-3, 'grad norms')
    assert_close(projs, g['grad_projs'], 2e-3, 'grad projs')


def test_network_state_dict_and_init(cuda):
    """Same seed -> same initial weights as the reference (digest captured by tools/gen_golden.py)."""
    from psnerf_amd.stage1 import NeuralNetwork
    g = np.load(os.path.join(GOLDEN, 'stage1_net_h256.npz'))
    torch.manual_seed(7)
    net = NeuralNetwork(stage1_cfg('bunny'))
    assert state_dict_digest(net.state_dict()) == str(g['init_digest'])
    assert sum(p.numel() for p in net.parameters()) == 802490
