"""
The geometry seams of tfrt/geometry.py, backed by the HIP kernels.

* ``snells_law_3D`` / ``snells_law_2D``   -> tfrt_snell3d / tfrt_snell2d (one ray per lane)
* ``line_triangle_intersect_nearest``     -> tfrt_intersect3d (the fused form of
  ``line_triangle_intersect`` + the nearest-hit reduction of engine.py:1132-1166; the dense
  (M, N) intermediate of the reference is never materialised)
* ``angle_in_interval``                   tiny comparison helper (geometry.py:766-802)

All of these require HIP tensors.
"""
import math

import torch

from . import ops

PI = math.pi


def snells_law_3D(x_start, y_start, z_start, x_end, y_end, z_end, norm, n_in, n_out,
                  new_ray_length):
    """geometry.py:671-753.  Returns (x_start, y_start, z_start, x_end, y_end, z_end) of the
    new rays (new start = old end)."""
    out = ops.snell3d(x_start, y_start, z_start, x_end, y_end, z_end, norm, n_in, n_out,
                      new_ray_length)
    return tuple(out[i] for i in range(6))


def snells_law_2D(x_start, y_start, x_end, y_end, norm, n_in, n_out, new_ray_length):
    """geometry.py:565-653.  Returns (x_start, y_start, x_end, y_end) of the new rays."""
    out = ops.snell2d(x_start, y_start, x_end, y_end, norm, n_in, n_out, new_ray_length)
    return tuple(out[i] for i in range(4))


def line_triangle_intersect_nearest(rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2,
                                    z2, epsilion, size_epsilion=1e-10, ray_start_epsilion=1e-10):
    rays = torch.stack([rx1, ry1, rz1, rx2, ry2, rz2])
    fv = torch.stack([xp, yp, zp, x1, y1, z1, x2, y2, z2], dim=1)
    return ops.intersect3d(rays, fv, epsilion, size_epsilion, ray_start_epsilion)


def angle_in_interval(angle, start, end):
    """geometry.py:766-802; inputs assumed in [-pi, pi]."""
    ra = angle - start
    ra = torch.where(ra < 0.0, ra + 2 * PI, ra)
    re = end - start
    re = torch.where(re < 0.0, re + 2 * PI, re)
    return ra <= re
