"""
Dispersion formulas, wavelength [nm] -> refractive index (tfrt/materials.py:23-104).

A material is a callable on a float64 torch tensor.  The trace evaluates every material once
per SOURCE ray (wavelength is inherited unchanged by child rays, operation.py:238-239) into a
small (n_materials, N) float64 table that the HIP kernels index by (material, source ray).
"""
import torch


def build_constant_material(n):
    return lambda x: n * torch.ones_like(x)


def acrylic(x):
    return torch.sqrt(
        2.1778 + 6.1209e-9 * x ** 2 - 1.5004e-15 * x ** 4 + 2.3678e4 * x ** -2
        - 4.2137e9 * x ** -4 + 7.3417e14 * x ** -6 - 4.5042e19 * x ** -8
    )


def _sellmeier(x, terms):
    x2 = x ** 2
    acc = 1
    for b, c in terms:
        acc = acc + b * x2 / (x2 - c)
    return torch.sqrt(acc)


def crown_glass(x):
    return _sellmeier(x, ((1.1273555e0, 7.20341707e3), (1.24412303e-1, 2.69835916e4),
                          (8.27100531e-1, 1.00384588e8)))


def flint_glass(x):
    return _sellmeier(x, ((1.34533359e0, 9.97743871e3), (2.09073176e-1, 4.70450767e4),
                          (9.37357162e-1, 1.11886764e8)))


def fused_silica(x):
    return _sellmeier(x, ((6.961663e-1, 4.679148e3), (4.079426e-1, 1.3512063e4),
                          (8.974794e-1, 9.7934002538e7)))


def polycarbonate(x):
    return _sellmeier(x, ((1.4182e0, 2.1304e4),))


def reflective(x):
    """n == 0 marks a mirror (geometry.py:747)."""
    return torch.zeros_like(x)


def soda_lime(x):
    return 1.5130e0 - 3.169e-9 * x ** 2 + 3.962e3 * x ** -2


def vacuum(x):
    return torch.ones_like(x)
