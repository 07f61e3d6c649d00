"""Checkpoint compatibility with the reference.

stage 1: one file {<module name>: state_dict, ..., epoch_it, it, loss_val_best}
         (stage1/model/checkpoints.py:29-41,97-111; train.py:66-73 registers model= and optimizer=).
stage 2: <checkpoints>/{ModelParameters,SGOptimizerParameters,SGSchedulerParameters,
         OptimizerLightParameters,LightParameters}/{N,latest}.pth (stage2/trainer.py:216-255, 171-197).
Because the modules keep the reference's state_dict keys, released checkpoints load unchanged.
"""
import os

import torch


class CheckpointIO(object):
    """stage1/model/checkpoints.py:9-111 (local files only; there is no network access here)."""

    def __init__(self, checkpoint_dir='./chkpts', **kwargs):
        self.module_dict = kwargs
        self.checkpoint_dir = checkpoint_dir
        os.makedirs(checkpoint_dir, exist_ok=True)

    def register_modules(self, **kwargs):
        self.module_dict.update(kwargs)

    def _path(self, filename):
        return filename if os.path.isabs(filename) else os.path.join(self.checkpoint_dir, filename)

    def save(self, filename, **kwargs):
        out = dict(kwargs)
        for k, v in self.module_dict.items():
            out[k] = v.state_dict()
        torch.save(out, self._path(filename))

    def load(self, filename, map_location=None):
        path = self._path(filename)
        if not os.path.exists(path):
            raise FileExistsError(path)  # the reference raises FileExistsError for a missing file (checkpoints.py:84)
        return self.parse_state_dict(torch.load(path, map_location=map_location))

    def parse_state_dict(self, state_dict):
        for k, v in self.module_dict.items():
            if k in state_dict:
                v.load_state_dict(state_dict[k])
            else:
                print('Warning: Could not find %s in checkpoint!' % k)
        return {k: v for k, v in state_dict.items() if k not in self.module_dict}


STAGE2_SUBDIRS = {
    'model': ('ModelParameters', 'model_state_dict'),
    'sg_optimizer': ('SGOptimizerParameters', 'optimizer_state_dict'),
    'sg_scheduler': ('SGSchedulerParameters', 'scheduler_state_dict'),
}


def save_stage2(step, checkpoints_path, epoch):
    """stage2/trainer.py:216-255 for a psnerf_amd.stage2.TrainStep."""
    objs = {'model': step.model, 'sg_optimizer': step.sg_optimizer, 'sg_scheduler': step.sg_scheduler}
    for name, (sub, key) in STAGE2_SUBDIRS.items():
        d = os.path.join(checkpoints_path, sub)
        os.makedirs(d, exist_ok=True)
        for fn in (str(epoch) + '.pth', 'latest.pth'):
            torch.save({'epoch': epoch, key: objs[name].state_dict()}, os.path.join(d, fn))
    d = os.path.join(checkpoints_path, 'OptimizerLightParameters')
    os.makedirs(d, exist_ok=True)
    for fn in (str(epoch) + '.pth', 'latest.pth'):
        torch.save({'epoch': epoch, 'optimizer_light_state_dict': step.light_optimizer.state_dict(),
                    'scheduler_light_state_dict': step.light_scheduler.state_dict() if getattr(step, 'light_decay', True) else None},
                   os.path.join(d, fn))  # (trainer.py:241: None without train.light_decay)
    d = os.path.join(checkpoints_path, 'LightParameters')
    os.makedirs(d, exist_ok=True)
    for fn in (str(epoch) + '.pth', 'latest.pth'):
        torch.save({'epoch': epoch, 'light_state_dict': step.light_para.state_dict(),
                    # (trainer.py:250,254: the table's state, or -- intensities not trained -- the model's scalar)
                    'light_inten_state_dict': step.light_inten_para.state_dict() if getattr(step, 'light_inten_train', True)
                    else step.model.light_int}, os.path.join(d, fn))


def load_stage2(step, checkpoints_path, checkpoint='latest', map_location=None):
    """stage2/trainer.py:171-197.  Returns the stored epoch."""
    def rd(sub):
        return torch.load(os.path.join(checkpoints_path, sub, str(checkpoint) + '.pth'), map_location=map_location)
    m = rd('ModelParameters')
    step.model.load_state_dict(m['model_state_dict'])
    step.sg_optimizer.load_state_dict(rd('SGOptimizerParameters')['optimizer_state_dict'])
    step.sg_scheduler.load_state_dict(rd('SGSchedulerParameters')['scheduler_state_dict'])
    if os.path.exists(os.path.join(checkpoints_path, 'OptimizerLightParameters', str(checkpoint) + '.pth')):
        d = rd('OptimizerLightParameters')
        step.light_optimizer.load_state_dict(d['optimizer_light_state_dict'])
        if d.get('scheduler_light_state_dict') is not None:
            step.light_scheduler.load_state_dict(d['scheduler_light_state_dict'])
        d = rd('LightParameters')
        step.light_para.load_state_dict(d['light_state_dict'])
        if isinstance(d['light_inten_state_dict'], dict):
            step.light_inten_para.load_state_dict(d['light_inten_state_dict'])
    return m['epoch']
