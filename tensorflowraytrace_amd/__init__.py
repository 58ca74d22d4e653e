"""
tensorflowraytrace_amd -- MI355X-native hot path of ecpoppenheimer/TensorFlowRayTrace (tfrt).

The package mirrors the reference's module layout for the ray-trace hot path
(``engine``, ``boundaries``, ``sources``, ``distributions``, ``operation``, ``materials``,
``optimizer``, ``update``, ``geometry``); tensors are torch (ROCm) tensors used as device
array containers, and everything on the per-step path -- ray x boundary intersection,
nearest hit, classification/compaction, Snell update, the pass loop and its reverse sweep --
runs in hand-written HIP kernels (``csrc/``) reached through the C ABI of
``include/tfrt_hip.h`` via ctypes.  There is no CPU fallback for that path.

``import tfrt`` (the shim package at the repo root) exposes the same modules under the
reference's names.
"""
from . import config  # noqa: F401
from .config import set_device, get_device, set_ray_dtype, get_ray_dtype  # noqa: F401

__version__ = "0.1.0"
