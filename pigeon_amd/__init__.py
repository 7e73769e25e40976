"""pigeon_amd — MI355X-native image-generation hot path of YiyanXu/Pigeon.

The package holds only what the path of Pigeon/models/lavit_for_pigeon.py:674-789 needs: the HIP
kernels behind a C ABI (csrc/, include/pigeon_hip.h) and the host-side mirror of the diffusers object
protocol the reference drives (UNet2DConditionModel / AttnProcessor / scheduler.step).
"""
__version__ = "0.1.0"
