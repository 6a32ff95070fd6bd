"""dfd-clip_amd: MI355X-native (gfx950) implementation of DFD-CLIP's hot path.

Per-clip CLIP-ViT key/value extraction plus the temporal cross-attention decoder, behind
the reference's own `Detector` model API (reference `src/models.py:394-780`).  Python host
code on PyTorch-ROCm (device memory, streams, `torch.distributed`) over a C-ABI shared
library of hand-written HIP kernels (`csrc/`, declared in `include/dfdclip.h`).
"""
from .config import ConfigNode, default_detector_config  # noqa: F401
from .weights import ARCHS, random_state_dict, synthetic_clips  # noqa: F401

__all__ = ["ConfigNode", "default_detector_config", "ARCHS", "random_state_dict", "synthetic_clips"]
