"""Process-wide defaults: which device holds scene tensors, and the ray-state dtype."""
import torch

_device = None
_ray_dtype = torch.float32


def get_device():
    """Device new scene tensors are created on (HIP device if one is visible, else CPU so
    that scenes can still be *built* -- tracing them needs the GPU)."""
    global _device
    if _device is None:
        _device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device("cpu")
    return _device


def set_device(device):
    global _device
    _device = torch.device(device)


def get_ray_dtype():
    """dtype of ray state inside the trace kernels (float32 default, float64 optional).
    Geometry, parameters, refractive indices and gradients are always float64."""
    return _ray_dtype


def set_ray_dtype(dtype):
    global _ray_dtype
    if dtype not in (torch.float32, torch.float64, torch.float16):
        raise ValueError("ray dtype must be torch.float32, torch.float64 or torch.float16")
    _ray_dtype = dtype


def as_f64(x, device=None):
    dev = device or get_device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float64)
    return torch.as_tensor(x, dtype=torch.float64, device=dev)
