"""
The geometry seams of tfrt/geometry.py, backed by the HIP kernels.

* ``snells_law_3D`` / ``snells_law_2D``   -> tfrt_snell3d / tfrt_snell2d (one ray per lane)
* ``line_triangle_intersect_nearest``     -> tfrt_intersect3d (the fused form of
  ``line_triangle_intersect`` + the nearest-hit reduction of engine.py:1132-1166; the dense
  (M, N) intermediate of the reference is never materialised)
* ``angle_in_interval``                   tiny comparison helper (geometry.py:766-802)
* ``line_intersect`` / ``raw_line_intersect``, ``line_triangle_intersect`` /
  ``raw_line_triangle_intersect``, ``line_circle_intersect`` / ``raw_line_circle_intersect``
  -> tfrt_line_intersect / tfrt_line_triangle_intersect / tfrt_line_circle_intersect: the dense
  pairwise functions with the reference's signatures and return values ((M, N) grids for the
  meshgrid forms, the operands' shape for the raw forms).  Forward only: the differentiable
  path is the fused trace (``OpticalEngine.ray_trace``).

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


# ------------------------------------------------------------ dense pairwise functions

def line_intersect(x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e, epsilion):
    """geometry.py:27-78: N first lines x M second lines -> x, y, valid, u, v of shape (M, N)
    (the reference's code returns this flat 5-tuple, geometry.py:167)."""
    return ops.line_intersect((x1s, y1s, x1e, y1e), (x2s, y2s, x2e, y2e), epsilion, grid=True)


def raw_line_intersect(x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e, epsilion):
    """geometry.py:96-167: element-wise on same-shaped operands.  Returns x, y, valid, u, v."""
    return ops.line_intersect((x1s, y1s, x1e, y1e), (x2s, y2s, x2e, y2e), epsilion, grid=False)


def line_triangle_intersect(rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2,
                            epsilion):
    """geometry.py:191-251: N rays x M triangles -> x, y, z, valid, ray_u, trig_u, trig_v (M, N)."""
    return ops.line_triangle_intersect((rx1, ry1, rz1, rx2, ry2, rz2),
                                       (xp, yp, zp, x1, y1, z1, x2, y2, z2), epsilion, grid=True)


def raw_line_triangle_intersect(rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2,
                                epsilion):
    """geometry.py:275-320: element-wise."""
    return ops.line_triangle_intersect((rx1, ry1, rz1, rx2, ry2, rz2),
                                       (xp, yp, zp, x1, y1, z1, x2, y2, z2), epsilion, grid=False)


def line_circle_intersect(xs, ys, xe, ye, xc, yc, r, epsilion):
    """geometry.py:338-402: N lines x M circles -> (plus, minus) dicts of (M, N) tensors
    ``x, y, valid, u, v`` (u: parameter along the line, v: angle of the hit on the circle)."""
    return ops.line_circle_intersect((xs, ys, xe, ye), (xc, yc, r), epsilion, grid=True)


def raw_line_circle_intersect(xs, ys, xe, ye, xc, yc, r, epsilion):
    """geometry.py:420-547: element-wise."""
    return ops.line_circle_intersect((xs, ys, xe, ye), (xc, yc, r), epsilion, grid=False)
