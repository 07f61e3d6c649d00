"""psnerf_amd -- MI355X-native (gfx950) hot path of PS-NeRF.

Layout:
  csrc/       hand-written HIP kernels + the C-ABI shared library (include/psnerf_hip.h)
  hip.py      ctypes binding of that C ABI (fails loudly when the .so is missing)
  ops.py      torch.autograd.Function wrappers (device pointers in, device pointers out)
  stage1/     NeuralNetwork / Renderer / Loss / Trainer with the reference's signatures
  stage2/     PSNetwork / SGBasis / MainLoss / NormalLoss / TrainStep with the reference's signatures
  dist.py     ray/pixel data-parallel helpers (RCCL via torch.distributed)
  synthetic.py  BEAR-shaped synthetic scenes for tests and bench

Importing this package does not touch the GPU or the shared library;
``psnerf_amd.hip`` does.
"""
__version__ = '0.1.0'
