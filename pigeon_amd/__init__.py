"""pigeon_amd — MI355X-native image-generation hot path of YiyanXu/Pigeon.

The package holds only what the path of Pigeon/models/lavit_for_pigeon.py:674-789 needs: the HIP
kernels behind a C ABI (csrc/, include/pigeon_hip.h) and the host-side mirror of the diffusers object
protocol the reference drives (UNet2DConditionModel / AttnProcessor / scheduler.step).
"""
__version__ = "0.1.0"

import os as _os

# MIOpen (3x3 convs of the UNet / VAE, reached through torch) otherwise benchmarks every solver the first time it sees
# a shape: ~80 s per model on a fresh machine.  Immediate-mode selection gives the same steady-state speed here.
_os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
_os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
