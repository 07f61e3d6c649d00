"""Stage-1 (UNISURF-style shape + appearance) with the reference's ``model`` package surface
(stage1/model/__init__.py:1-4)."""
from ..checkpoints import CheckpointIO
from .network import NeuralNetwork, WNLinear
from .rendering import Renderer
from .losses import Loss
from .training import Trainer
from . import config
