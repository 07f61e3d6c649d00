"""Engine selection is loud: under ``ops.STRICT`` a device tensor that falls off the register-resident /
fused engines raises instead of quietly taking a layer-wise or torch formulation, and one train step of EVERY shipped
configuration (the seven objects of stage1/configs/*.yaml and stage2/confs/*.conf, tests/golden/configs.json) lands on them."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, stage1_cfg
from psnerf_amd.synthetic import stage1_camera, stage2_inputs

pytestmark = pytest.mark.gpu
OBJECTS = sorted(json.load(open(os.path.join(GOLDEN, 'configs.json')))['stage1'])


def test_the_fixture_lists_the_seven_objects():
    assert OBJECTS == ['armadillo', 'bear', 'buddha', 'bunny', 'cow', 'pot2', 'reading']


@pytest.mark.parametrize('obj', OBJECTS)
def test_stage1_train_step_of_every_shipped_config_takes_the_fused_engines(cuda, obj):
    """stage1/train.py's step (Trainer.train_step: march sweep + root finder, sampling, geometry chains with the hand-derived
    double backward, appearance chains, composite, losses, Adam) on the object's own yaml values, iterations 0 and 6000 (the
    64 -> 96 + 32 sample switch is not in the yaml but `it` moves the interval)."""
    from psnerf_amd import ops
    from psnerf_amd.optim import FlatAdam
    from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
    cfg = stage1_cfg(obj, **{'training.n_training_points': 256})
    torch.manual_seed(3)
    net = NeuralNetwork(cfg)
    ren = Renderer(net, cfg, device=cuda)
    tr = Trainer(ren, FlatAdam(net.parameters(), lr=1e-4), cfg, device=cuda)
    h, w = 40, 48
    K, c2w, S = stage1_camera(cfg, h=h, w=w)
    g = torch.Generator().manual_seed(1)
    batch = {'img': torch.rand(1, 3, h, w, generator=g).to(cuda), 'img.mask': (torch.rand(1, h, w, generator=g) > 0.3).float().to(cuda),
             'img.world_mat': c2w.to(cuda), 'img.camera_mat': K.to(cuda), 'img.scale_mat': S.to(cuda),
             'img.normal': torch.nn.functional.normalize(torch.randn(1, 3, h, w, generator=g), dim=1).to(cuda),
             'img.norm_mask': (torch.rand(1, h, w, generator=g) > 0.5).float().to(cuda)}
    ops.reset_hits()
    with ops.strict():
        for it in (0, 6000):
            terms = tr.train_step(batch, it=it)
    torch.cuda.synchronize()
    assert np.isfinite(float(terms['loss']))
    assert not ops.FALLBACKS, dict(ops.FALLBACKS)
    for name in ('GeoFieldFused', 'AppNetFused', 'AlphaComposite', 'march_sweep', 'root_find'):
        assert ops.HITS[name] >= 2, (name, dict(ops.HITS))
    assert ops.HITS['GeoField'] == 0 and ops.HITS['ReluMLP'] == 0, dict(ops.HITS)


@pytest.mark.parametrize('obj', OBJECTS)
def test_stage2_train_step_of_every_shipped_config_takes_the_fused_engines(cuda, obj):
    """stage2/trainer.py:355-410 on the object's own conf values (light_bs 10, V = 8; brdf.light_intensity and the intensity-table
    switch differ per object), one step in each train_fix phase."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd import ops
    from psnerf_amd.stage2.conf import object_conf
    conf = object_conf(obj)
    torch.manual_seed(5)
    net = s2.PSNetwork(conf).to(cuda)
    NL = 24
    light_init = torch.nn.functional.normalize(torch.randn(NL, 3), dim=-1)
    step = s2.TrainStep(net, conf, NL, light_init.to(cuda), cuda)
    L, V = conf.get_int('train.light_bs'), conf.get_int('train.vis_train_num')
    inp, gt = stage2_inputs(1500, L, V, seed=2)
    inp = {k: v.to(cuda) for k, v in inp.items()}
    gt = {k: v.to(cuda) for k, v in gt.items()}
    l_slt = torch.arange(L, device=cuda)
    ops.reset_hits()
    with ops.strict():
        for it in (10, 5001):
            step.cur_iter = it
            terms, out = step.step(inp, gt, l_slt, train_order=True)
    torch.cuda.synchronize()
    assert np.isfinite(float(terms['total']))
    assert not ops.FALLBACKS, dict(ops.FALLBACKS)
    for name in ('VisibilityPair', 'FusedReluNet', 'SGShade', 'Stage2Losses'):
        assert ops.HITS[name] >= 2, (name, dict(ops.HITS))
    assert ops.HITS['ReluMLP'] == 0 and ops.HITS['FusedPairMLP'] == 0, dict(ops.HITS)


def test_strict_raises_where_the_default_falls_back(cuda):
    """A width the register-resident engines do not cover (192): by default the layer-wise HIP GEMMs run and the event is
    counted; under ops.strict() the same call raises.  CPU tensors never count (ops.fallback ignores them)."""
    from psnerf_amd import ops
    from psnerf_amd.stage2.renderer import MLP
    torch.manual_seed(0)
    m = MLP(39, 3, 192, 4, skip_at=(2,), final='sigmoid').to(cuda)
    cols = torch.arange(39, device=cuda)
    x = torch.randn(100, 64, device=cuda)
    ops.reset_hits()
    with ops.strict():
        ops.fallback('a CPU tensor', torch.zeros(3), 'not a fallback')
    assert not ops.FALLBACKS
    h0 = ops.HITS['ReluMLP']
    y = m(x, cols)
    assert sum(ops.FALLBACKS.values()) == 1 and ops.HITS['ReluMLP'] == h0 + 1 and y.shape == (100, 3)
    with ops.strict(), pytest.raises(RuntimeError, match='STRICT'):
        m(x, cols)
    # a deliberate switch is not a fallback
    m.FUSED = False
    with ops.strict():
        y2 = m(x, cols)
    assert torch.equal(y, y2)
