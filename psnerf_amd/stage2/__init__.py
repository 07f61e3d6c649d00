"""Stage-2 (SVBRDF + light joint optimisation) with the reference's module surface."""
from .conf import Conf, bear_conf, load_conf, parse_conf
from .loss import MainLoss, NormalLoss
from .renderer import MLP, PSNetwork, SGBasis
from .trainer import TrainStep, psnr
