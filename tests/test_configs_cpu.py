"""Configuration readers and the module-level drop-in surface, on the CPU.

tests/golden/configs.json holds the hot-path VALUES of all 14 reference configuration files (7 x stage1/configs/*.yaml read by
the reference's own loader, 7 x stage2/confs/*.conf read by an independent scanner in tools/gen_golden.py -- pyhocon is not
in this image); the build's readers, ``stage1_cfg`` / ``bear_conf`` / ``object_conf`` and the INTEGRATION.md section A lines
are checked against it."""
import json
import os

import pytest
import torch

from tests.helpers import GOLDEN, ROOT

CFG = json.load(open(os.path.join(GOLDEN, 'configs.json')))
OBJECTS = ('armadillo', 'bear', 'buddha', 'bunny', 'cow', 'pot2', 'reading')


def test_fixture_covers_all_reference_objects():
    assert sorted(CFG['stage1']) == sorted(OBJECTS) and sorted(CFG['stage2']) == sorted(OBJECTS)
    # the seven stage-1 configs differ in the depth range only; the values VERDICT r3 quoted
    assert {o: (CFG['stage1'][o]['rendering']['near'], CFG['stage1'][o]['rendering']['far']) for o in OBJECTS} == {
        'armadillo': (2, 6), 'bear': (28, 35), 'buddha': (20, 30), 'bunny': (2, 6), 'cow': (32, 39), 'pot2': (23, 30),
        'reading': (33, 42)}


@pytest.mark.parametrize('obj', OBJECTS)
def test_stage1_cfg_matches_reference_yaml(obj):
    """synthetic.stage1_cfg(obj) == the hot-path values of stage1/configs/<obj>.yaml (every key the modules read)."""
    from psnerf_amd.synthetic import stage1_cfg
    cfg = stage1_cfg(obj)
    for sec, keys in CFG['stage1'][obj].items():
        for k, v in keys.items():
            assert cfg[sec][k] == v, (obj, sec, k, cfg[sec].get(k), v)


def test_stage1_cfg_raises_on_unknown_object():
    from psnerf_amd.synthetic import stage1_cfg
    with pytest.raises(ValueError, match='unknown object'):
        stage1_cfg('teapot')
    from psnerf_amd.stage2.conf import object_conf
    with pytest.raises(ValueError, match='unknown object'):
        object_conf('teapot')


def test_yaml_reader_inherit_from_and_defaults(tmp_path):
    """stage1/dataloading/configloading.py:3-49 semantics: inherit_from chains, default_path at the end of the chain,
    nested dictionaries merged key by key, scalars / lists replaced."""
    from psnerf_amd.stage1.config import hot_path, load_config
    (tmp_path / 'default.yaml').write_text('model:\n  hidden_dim: 64\n  skips: [2]\nrendering:\n  near: 1\n  far: 2\n  radius: 2.0\n')
    (tmp_path / 'base.yaml').write_text('rendering:\n  near: 28\n  type: unisurf\ntraining:\n  n_training_points: 2048\n')
    (tmp_path / 'obj.yaml').write_text('inherit_from: %s\nmodel:\n  skips: [4]\nrendering:\n  far: 35\n' % (tmp_path / 'base.yaml'))
    cfg = load_config(str(tmp_path / 'obj.yaml'), str(tmp_path / 'default.yaml'))
    assert cfg['model'] == {'hidden_dim': 64, 'skips': [4]}
    assert cfg['rendering'] == {'near': 28, 'far': 35, 'radius': 2.0, 'type': 'unisurf'}
    assert cfg['training'] == {'n_training_points': 2048}
    assert cfg['inherit_from'] == str(tmp_path / 'base.yaml')
    assert load_config(str(tmp_path / 'base.yaml'))['rendering'] == {'near': 28, 'type': 'unisurf'}
    assert hot_path(cfg)['rendering'] == {'type': 'unisurf', 'near': 28, 'far': 35, 'radius': 2.0}


@pytest.mark.parametrize('obj', OBJECTS)
def test_yaml_reader_roundtrip_of_the_fixture(obj, tmp_path):
    """A YAML file carrying the fixture's values reads back to the same dictionary, and builds the same network as stage1_cfg."""
    import yaml
    from psnerf_amd.stage1.config import hot_path, load_config
    from psnerf_amd.synthetic import stage1_cfg
    path = tmp_path / ('%s.yaml' % obj)
    path.write_text(yaml.safe_dump(CFG['stage1'][obj]))
    cfg = load_config(str(path))
    assert hot_path(cfg) == CFG['stage1'][obj]
    ref = stage1_cfg(obj)
    for sec in ('model', 'rendering'):
        for k, v in cfg[sec].items():
            assert ref[sec][k] == v


@pytest.mark.parametrize('obj', OBJECTS)
def test_stage2_conf_reader_and_object_conf(obj, tmp_path):
    """parse_conf on a .conf text carrying the fixture's values returns them through the pyhocon-style accessors, and
    ``object_conf(obj)`` (``bear_conf()`` for bear) agrees with the fixture on every key it defines."""
    from psnerf_amd.stage2.conf import bear_conf, object_conf, parse_conf
    flat = CFG['stage2'][obj]
    tree = {}
    for k, v in flat.items():
        node = tree
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v

    def emit(node, ind=0):
        lines = []
        for k, v in node.items():
            if isinstance(v, dict):
                lines += [' ' * ind + '%s{' % k] + emit(v, ind + 4) + [' ' * ind + '}']
            elif isinstance(v, list):
                lines.append(' ' * ind + '%s = [%s]  # a comment' % (k, ','.join(str(x) for x in v)))
            else:
                lines.append(' ' * ind + '%s = %s' % (k, v))
        return lines
    conf = parse_conf('\n'.join(emit(tree)))
    for k, v in flat.items():
        if isinstance(v, bool):
            assert conf.get_bool(k) is v, k
        elif isinstance(v, int):
            assert conf.get_int(k) == v, k
        elif isinstance(v, float):
            assert conf.get_float(k) == v, k
        elif isinstance(v, list):
            assert conf.get_list(k) == v, k
        else:
            assert conf.get_string(k) == v, k
    assert conf.get_int('train.not_there', default=7) == 7
    with pytest.raises(KeyError):
        conf.get_int('train.not_there')
    built = object_conf(obj)
    if obj == 'bear':
        assert built == bear_conf()

    def walk(node, prefix=''):
        for k, v in node.items():
            if isinstance(v, dict):
                yield from walk(v, prefix + k + '.')
            else:
                yield prefix + k, v
    n = 0
    for k, v in walk(built):
        if k in flat:
            assert flat[k] == v, (obj, k, flat[k], v)
            n += 1
        else:
            # keys the reference reads with a default that the shipped file leaves out
            assert (obj in ('bunny', 'armadillo') and k == 'train.light_inten_train') or k in ('brdf.fresnel_f0',), (obj, k)
    assert n >= 40


def test_integration_section_a_lines_run_against_psnerf_amd(tmp_path):
    """INTEGRATION.md section A, executed: the import / constructor lines of stage1/train.py:9-10,56-72 with
    ``import psnerf_amd.stage1 as mdl`` and the dotted-name resolution of stage2/trainer.py:109-113 (utils.get_class,
    stage2/utils/general.py:9-15) with ``psnerf_amd.stage2...`` class names."""
    import torch.optim as optim
    import psnerf_amd.stage1 as mdl
    from psnerf_amd.synthetic import stage1_batch, stage1_cfg
    cfg = stage1_cfg('bear', **{'model.hidden_dim': 64, 'model.feat_size': 64})
    device = torch.device('cpu')
    model = mdl.NeuralNetwork(cfg)                                                   # train.py:57
    renderer = mdl.Renderer(model, cfg, device=device)                               # :60
    optimizer = optim.Adam(model.parameters(), lr=cfg['training']['learning_rate'],
                           weight_decay=cfg['training']['weight_decay'])            # :62-63
    trainer = mdl.Trainer(renderer, optimizer, cfg, device=device)                   # :65
    checkpoint_io = mdl.CheckpointIO(str(tmp_path / 'models'), model=model, optimizer=optimizer)   # :66
    try:
        load_dict = checkpoint_io.load('model.pt')                                   # :68-71
    except FileExistsError:
        load_dict = dict()
    assert load_dict.get('epoch_it', -1) == -1 and load_dict.get('it', -1) == -1
    scheduler = optim.lr_scheduler.MultiStepLR(optimizer, cfg['training']['scheduler_milestones'],
                                               gamma=cfg['training']['scheduler_gamma'], last_epoch=-1)   # :75-77
    assert scheduler.get_last_lr() == [1e-4]
    checkpoint_io.save('model.pt', epoch_it=3, it=77, loss_val_best=0.5)             # :120-122
    assert checkpoint_io.load('model.pt')['it'] == 77
    # training.py:120-139 -- the data dictionary of the loader -> the nine tensors of the reference order
    data = stage1_batch(cfg, h=8, w=10, seed=0)
    data['img.idx'] = torch.tensor([0])
    img, mask_img, world_mat, camera_mat, scale_mat, img_idx, normal, norm_mask, mask_valid = trainer.process_data_dict(data)
    assert img.shape == (1, 3, 8, 10) and mask_img.shape == (1, 1, 8, 10) and norm_mask.shape == (1, 1, 8, 10)
    assert mask_valid.shape == (1, 1, 8, 10) and bool((mask_valid == 1).all()) and normal.shape == (1, 3, 8, 10)
    assert world_mat.shape == camera_mat.shape == scale_mat.shape == (1, 4, 4) and int(img_idx) == 0
    assert all(hasattr(model, a) for a in ('rescale', 'skips', 'octaves_pe'))        # attributes callers read

    def get_class(kls):                                                              # stage2/utils/general.py:9-15
        parts = kls.split('.')
        module = '.'.join(parts[:-1])
        m = __import__(module)
        for comp in parts[1:]:
            m = getattr(m, comp)
        return m
    import psnerf_amd.stage2 as s2
    conf = s2.bear_conf()
    net = get_class('psnerf_amd.stage2.renderer.PSNetwork')(conf=conf)               # trainer.py:109
    loss = get_class('psnerf_amd.stage2.loss.MainLoss')(**conf.get_config('loss'))   # :111
    loss_n = get_class('psnerf_amd.stage2.loss.NormalLoss')(**conf.get_config('normal.loss'))   # :113
    assert isinstance(net, s2.PSNetwork) and loss.vis_weight == 1 and loss_n.normal_smooth_weight == 0.05
    sg_optimizer = torch.optim.Adam(net.parameters(), lr=conf.get_float('train.sg_learning_rate'))   # :115-116
    assert len(sg_optimizer.param_groups[0]['params']) == len(list(net.parameters()))


def test_bench_watchdog_prints_the_headline_when_the_diagnostic_object_hangs():
    """bench.guarded: strong_cfg4 at N > 1 (HIP graphs around RCCL collectives) runs behind a watchdog -- if it never returns, rank 0
    prints the line it had assembled, with an error in place of the object, and the process leaves with exit code 3 (a hung run is not a success)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.guarded(lambda: time.sleep(60), {'metric': 'm', 'value': 1.5, 'strong_cfg4': None}, 0, timeout=1)") % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=50)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] == 1.5 and 'did not return' in line['strong_cfg4']['error']
    # a rank other than 0 leaves silently; a function that returns in time is passed through
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench; "
             "print(bench.guarded(lambda: 7, {'x': 1}, 0, timeout=5), flush=True); bench.guarded(lambda: time.sleep(60), {'x': 1}, 1, timeout=1)") % root
    r = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, timeout=50)
    assert r.returncode == 3 and r.stdout.strip() == '7', (r.stdout, r.stderr[-300:])


def test_train_step_reads_the_trainer_switches_with_the_reference_defaults():
    """stage2/trainer.py:36-50: train.light_train / train.ana_fixlight / train.visibility / train.vis_loss default to False in the
    reference; a configuration that leaves them out must train what the reference would (no light tables, no visibility loss).
    The switches are implemented (goldens: tests/golden/stage2_trainer_{gtlight,fixlight,novisloss}.npz); train.vis_plus without
    train.light_train fails like the reference (trainer.py:149,388: 'light_vis_train' does not exist there).  The two switches that
    vary between the shipped configurations -- train.light_inten_train (off for bunny / armadillo) and train.light_decay -- are read."""
    import torch
    import psnerf_amd.stage2 as s2
    li = torch.nn.functional.normalize(torch.randn(6, 3), dim=-1)
    conf = s2.bear_conf()
    for k in ('light_train', 'vis_loss', 'ana_fixlight'):
        conf['train'].pop(k, None)
    st = s2.TrainStep(s2.PSNetwork(conf), conf, 6, li, torch.device('cpu'))
    assert not st.light_train and not st.vis_loss and st.visibility and not st.ana_fixlight and not st.light_inten_train
    assert not st.light_para.weight.requires_grad
    st.train_fix()   # iteration 0: without the loss the visibility net is frozen (trainer.py:498-499) ...
    assert not any(q.requires_grad for q in st.model.visibility_net.parameters())
    st.cur_iter = 5000
    st.train_fix()   # ... and never released; the tables of a run without light_train stay frozen as well
    assert not any(q.requires_grad for q in st.model.visibility_net.parameters()) and not st.light_para.weight.requires_grad
    conf = s2.bear_conf(**{'train.ana_fixlight': True})
    st = s2.TrainStep(s2.PSNetwork(conf), conf, 6, li, torch.device('cpu'))
    st.train_fix(); st.cur_iter = 5000; st.train_fix()
    assert st.light_train and not st.light_para.weight.requires_grad and all(q.requires_grad for q in st.model.albedo_net.parameters())
    conf = s2.bear_conf(**{'train.light_train': False})
    with pytest.raises(AttributeError):
        s2.TrainStep(s2.PSNetwork(conf), conf, 6, li, torch.device('cpu'), vis_plus=object())
    for obj, inten in (('bear', True), ('reading', True), ('armadillo', False), ('bunny', False)):
        from psnerf_amd.stage2.conf import object_conf
        conf = object_conf(obj)
        st = s2.TrainStep(s2.PSNetwork(conf), conf, 6, li, torch.device('cpu'))
        assert st.light_inten_train == inten and st.light_decay and len(st.light_optimizer.param_groups) == (2 if inten else 1), obj
        assert st.light_inten_para.weight.requires_grad == inten


def test_scheduler_milestones_follow_the_reference_scaling():
    """stage2/trainer.py:118-121: milestones = epochs x len(dataset) (x light_bs under multi_light)."""
    import psnerf_amd.stage2 as s2
    from psnerf_amd.stage2.trainer import scheduler_milestones
    conf = s2.bear_conf()
    assert scheduler_milestones(conf, 20) == [m * 20 * 10 for m in (200, 400, 600, 800, 1000)]
    assert scheduler_milestones(s2.bear_conf(**{'train.multi_light': False}), 20) == [m * 20 for m in (200, 400, 600, 800, 1000)]
    assert scheduler_milestones(s2.bear_conf(**{'train.sg_sched_milestones': []}), 20) == []


def test_vis_plus_directions_are_spread_unit_vectors_facing_the_camera():
    """handoff.sample_vis_plus_dirs (stage1/shape_extract.py:117-129): unit vectors, the requested count, hemisphere filter,
    farthest-point spread (the closest pair is far wider apart than in a random subset of the same size)."""
    import numpy as np
    import torch
    from psnerf_amd import handoff
    wm = torch.eye(4)[None]
    d = handoff.sample_vis_plus_dirs(wm, vnum=64, semisphere=True, rng=np.random.RandomState(3))
    assert d.shape == (64, 3) and d.dtype == torch.float32
    assert torch.allclose(d.norm(dim=-1), torch.ones(64), atol=1e-6)
    assert bool((d[:, 2] < 0).all())  # view_dir = world_mat[0, :3, 2] = +z: only directions with a negative z component
    gram = d @ d.t() - 2 * torch.eye(64)
    rs = np.random.RandomState(4)
    v = rs.normal(size=(2000, 3)); v = v / np.linalg.norm(v, axis=-1, keepdims=True); v = v[v[:, 2] < 0][:64]
    gram_r = torch.from_numpy(v @ v.T).float() - 2 * torch.eye(64)
    assert float(gram.max()) < float(gram_r.max()) - 0.02   # largest cosine between two different directions: smaller = better spread
    full = handoff.sample_vis_plus_dirs(wm, vnum=256, rng=np.random.RandomState(5))
    assert full.shape == (256, 3) and bool((full[:, 2] > 0).any()) and bool((full[:, 2] < 0).any())
