"""Shared test helpers: deterministic weights, gradient digests, tolerances."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

from psnerf_amd.synthetic import perturb_state_dict, state_dict_digest, stage1_cfg  # noqa: E402


def stage1_state_dict(cfg, seed):
    """Deterministic, non-degenerate stage-1 weights (oracle init + seeded jitter)."""
    from oracle import stage1 as o1
    torch.manual_seed(seed)
    net = o1.NeuralNetwork(cfg)
    return perturb_state_dict(net.state_dict(), seed + 1)


def stage2_state_dict(conf, seed):
    from oracle import stage2 as o2
    torch.manual_seed(seed)
    net = o2.PSNetwork(conf)
    sd = net.state_dict()
    # default nn.Linear init leaves the final visibility/albedo outputs tiny; jitter biases a bit
    return perturb_state_dict(sd, seed + 1, scale=0.05)


def grad_digest(named_grads, seed=1234):
    """Compact fingerprint of a set of gradients: per-tensor L2 norm and a dot
    product with a fixed seeded random vector (sensitive to any element)."""
    g = torch.Generator().manual_seed(seed)
    names, norms, projs = [], [], []
    for name in sorted(named_grads.keys()):
        v = named_grads[name]
        if v is None:
            continue
        v = v.detach().double().cpu().reshape(-1)
        r = torch.randn(v.numel(), generator=g, dtype=torch.float64)
        names.append(name)
        norms.append(float(v.norm()))
        projs.append(float((v * r).sum()))
    return names, np.array(norms), np.array(projs)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def assert_close(a, b, rtol, name=''):
    e = rel_err(a, b)
    assert e <= rtol, '%s: max-abs err / max-abs ref = %.3e > %.1e' % (name, e, rtol)
