"""Minimal HOCON-subset reader for the reference's stage2/confs/*.conf (pyhocon is not a dependency).

The shipped conf files only use ``section{ ... }`` nesting, ``key = value`` pairs, ``#`` comments and
flat lists; values are ints, floats, booleans, bare or quoted strings.  ``Conf`` exposes the accessor
subset the reference calls on the pyhocon tree: get_string / get_int / get_float / get_bool / get_list
with dotted keys and ``default=`` (stage2/model/renderer.py:60-108, stage2/trainer.py:25-57).
"""
import re

_MISSING = object()


class Conf(dict):
    def _get(self, key, default):
        node = self
        for part in key.split('.'):
            if not isinstance(node, dict) or part not in node:
                if default is _MISSING:
                    raise KeyError('No configuration setting found for key %s' % key)
                return default
            node = node[part]
        return node

    def get_string(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if v is None else str(v)

    def get_int(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if v is None else int(v)

    def get_float(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if v is None else float(v)

    def get_bool(self, key, default=_MISSING):
        v = self._get(key, default)
        if isinstance(v, str):
            return v.lower() in ('true', 'yes', 'on')
        return bool(v)

    def get_list(self, key, default=_MISSING):
        return list(self._get(key, default))

    def get_config(self, key, default=_MISSING):
        return self._get(key, default)


def _value(tok):
    tok = tok.strip()
    if tok.startswith('[') and tok.endswith(']'):
        inner = tok[1:-1].strip()
        return [_value(t) for t in inner.split(',')] if inner else []
    if len(tok) >= 2 and tok[0] == tok[-1] and tok[0] in '"\'':
        return tok[1:-1]
    low = tok.lower()
    if low in ('true', 'false'):
        return low == 'true'
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def parse_conf(text):
    root = Conf()
    stack = [root]
    for raw in text.splitlines():
        line = re.sub(r'(#|//).*$', '', raw).strip()
        while line:
            if line.startswith('}'):
                stack.pop()
                line = line[1:].strip()
                continue
            m = re.match(r'^([A-Za-z0-9_.\-]+)\s*\{(.*)$', line)
            if m:
                child = Conf()
                stack[-1][m.group(1)] = child
                stack.append(child)
                line = m.group(2).strip()
                continue
            m = re.match(r'^([A-Za-z0-9_.\-]+)\s*[=:]\s*(.*?)(\}?)$', line)
            if not m:
                raise ValueError('cannot parse conf line: %r' % raw)
            stack[-1][m.group(1)] = _value(m.group(2))
            line = m.group(3)
    return root


def load_conf(path):
    with open(path, 'r') as f:
        return parse_conf(f.read())


# brdf.light_intensity per object (stage2/confs/<obj>.conf; the synthetic objects use 4.0 and leave the per-light
# intensities fixed): the only hot-path values in which the seven reference confs differ
STAGE2_LIGHT_INTENSITY = {'bear': 2.0, 'buddha': 2.0, 'cow': 2.0, 'pot2': 2.0, 'reading': 2.0, 'bunny': 4.0, 'armadillo': 4.0}


def object_conf(obj, **overrides):
    """The hot-path subset of stage2/confs/<obj>.conf; an unknown object raises (load_conf reads the files themselves)."""
    if obj not in STAGE2_LIGHT_INTENSITY:
        raise ValueError('object_conf: unknown object %r (known: %s)' % (obj, ', '.join(sorted(STAGE2_LIGHT_INTENSITY))))
    c = bear_conf(**{'brdf.light_intensity': STAGE2_LIGHT_INTENSITY[obj]})
    c['train']['light_inten_train'] = obj not in ('bunny', 'armadillo')
    return _override(c, overrides)


def _override(c, overrides):
    for k, v in overrides.items():
        node = c
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, Conf())
        node[parts[-1]] = v
    return c


def bear_conf(**overrides):
    """The hot-path subset of stage2/confs/bear.conf (identical for every object except paths and
    brdf.light_intensity).  ``overrides`` use dotted keys."""
    c = Conf({
        'train': Conf(render_model='sgbasis', nbasis=9, specular_rgb=True, visibility=True, vis_loss=True,
                      light_vis_detach=True, vis_rgb_detach=True, normal_mlp=True, normal_joint=True,
                      shape_pregen=True, light_bs=10, vis_train_num=8, vis_plus=True, train_order=True, light_train=True,
                      multi_light=True, light_inten_train=True, light_decay=True, num_pixels=8192, sg_learning_rate=5e-4,
                      sg_sched_milestones=[200, 400, 600, 800, 1000],
                      light_learning_rate=5e-4, light_inten_lr=1e-3, sg_sched_factor=0.5),
        'loss': Conf(sg_rgb_weight=1.0, loss_type='L1', albedo_smooth_weight=0.05, rough_smooth_weight=0.01,
                     vis_weight=1),
        'brdf': Conf(net=Conf(n_freqs_xyz=10, mlp_width=128, mlp_depth=4, mlp_skip_at=2, xyz_jitter_std=0.01),
                     sgnet=Conf(mlp_width=64, mlp_depth=2, mlp_skip_at=-1), fresnel_f0=0.05, light_intensity=2.0),
        'normal': Conf(net=Conf(n_freqs_xyz=10, mlp_width=128, mlp_depth=4, mlp_skip_at=2, xyz_jitter_std=0.0),
                       loss=Conf(normal_weight=1, normal_smooth_weight=0.05)),
        'visibility': Conf(net=Conf(n_freqs_xyz=10, mlp_width=256, mlp_depth=8, mlp_skip_at=4)),
    })
    for k, v in overrides.items():
        node = c
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, Conf())
        node[parts[-1]] = v
    return c
