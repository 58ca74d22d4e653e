"""
``Variable``: the stand-in for ``tf.Variable`` -- a float64 leaf tensor that autograd tracks,
with the handful of in-place methods the reference calls on its parameters
(``assign``, ``assign_add``, ``assign_sub``, ``numpy``; boundaries.py:156,213,230,1614,
optimizer.py:282, dev/hexalens.py:335-336).
"""
import numpy as np
import torch

from . import config


class Variable(torch.nn.Parameter):
    def __new__(cls, data, dtype=torch.float64, device=None, requires_grad=True):
        t = torch.as_tensor(np.asarray(data) if not isinstance(data, torch.Tensor) else data)
        t = t.detach().to(dtype=dtype, device=device or config.get_device()).clone()
        return torch.nn.Parameter.__new__(cls, t, requires_grad)

    def assign(self, value):
        with torch.no_grad():
            self.copy_(torch.as_tensor(value, dtype=self.dtype, device=self.device))
        return self

    def assign_add(self, value):
        with torch.no_grad():
            self.add_(torch.as_tensor(value, dtype=self.dtype, device=self.device))
        return self

    def assign_sub(self, value):
        with torch.no_grad():
            self.sub_(torch.as_tensor(value, dtype=self.dtype, device=self.device))
        return self

    def numpy(self):
        return self.detach().cpu().numpy()

    def __deepcopy__(self, memo):
        return Variable(self.detach().clone(), dtype=self.dtype, device=self.device,
                        requires_grad=self.requires_grad)
