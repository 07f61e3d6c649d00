"""Stage-1 configuration files: the reader of stage1/dataloading/configloading.py:3-49 (YAML with ``inherit_from``
chains and an optional default file, later files overriding earlier ones key by key), so that the reference's
``stage1/configs/*.yaml`` load unchanged:

    cfg = psnerf_amd.stage1.config.load_config('configs/bear.yaml')        # stage1/train.py:30
    model = psnerf_amd.stage1.NeuralNetwork(cfg)

Only PyYAML is needed (the reference imports the same package).  ``hot_path(cfg)`` extracts the values the accelerated
path reads -- what tests/golden/configs.json pins for all seven reference objects."""
import yaml

# keys of the three sections the hot path reads (stage1/model/network.py:14-27, rendering.py:17-26, training.py:22-44)
HOT_KEYS = {
    'model': ('num_layers', 'hidden_dim', 'octaves_pe', 'octaves_pe_views', 'skips', 'geometric_init', 'feat_size', 'rescale'),
    'rendering': ('type', 'n_max_network_queries', 'white_background', 'near', 'far', 'radius', 'interval_start',
                  'interval_end', 'interval_decay', 'num_points_in', 'num_points_out', 'ray_marching_steps', 'occ_prob_points'),
    'training': ('type', 'normal_loss', 'normal_after', 'normal_angle', 'lambda_normloss', 'lambda_mask', 'mask_loss',
                 'n_training_points', 'learning_rate', 'weight_decay', 'scheduler_milestones', 'scheduler_gamma',
                 'lambda_l1_rgb', 'lambda_normals'),
}


def update_recursive(dst, src):
    """configloading.py:35-49: nested dictionaries are merged, everything else is replaced."""
    for k, v in src.items():
        if k not in dst:
            dst[k] = dict()
        if isinstance(v, dict):
            update_recursive(dst[k], v)
        else:
            dst[k] = v


def load_config(path, default_path=None):
    """configloading.py:3-32: the file itself, on top of the file it names under ``inherit_from`` (recursively), on top of
    ``default_path`` when the chain ends without one."""
    with open(path, 'r') as f:
        special = yaml.safe_load(f) or {}
    parent = special.get('inherit_from')
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, 'r') as f:
            cfg = yaml.safe_load(f) or {}
    else:
        cfg = dict()
    update_recursive(cfg, special)
    return cfg


def hot_path(cfg):
    """{section: {key: value}} of the keys the accelerated path reads (absent keys are left out: the modules apply the
    reference's ``cfg.get`` defaults)."""
    return {sec: {k: cfg[sec][k] for k in keys if k in cfg.get(sec, {})} for sec, keys in HOT_KEYS.items()}
