"""
CPU float64 restatement of tfrt/geometry.py (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function follows the reference's operation order so that rounding behaviour and the
"safe value" masking (which controls where gradients are zero) are the same.  TensorFlow
semantics that matter and their torch spellings:

* ``tf.meshgrid(a, b)`` (default 'xy' indexing) gives shape (len(b), len(a)): boundaries
  vary along rows, rays along columns.  Spelled here with broadcasting
  ``a[None, :]`` / ``b[:, None]``.
* ``tf.math.mod`` is floor-mod -> ``torch.remainder``.
* ``tf.math.l2_normalize(x, axis)`` = ``x * rsqrt(max(sum(x**2), 1e-12))``.
* ``tf.where`` passes zero gradient to the unselected branch -> ``torch.where`` does too.
* ``tf.sqrt`` on the CPU is Eigen's hardware square root (sqrtpd): correctly rounded.  This
  image's ``torch.sqrt`` is not (MKL VML / Sleef "u05": about 0.9 % of float64 results are one ulp
  off the correctly rounded value, measured against numpy), so the restatement uses ``sqrt``
  below (numpy's IEEE square root with torch's own derivative).  ``torch.rsqrt``, products, sums
  and quotients were checked bit for bit against numpy on 1M operands and are IEEE here.
"""
import math

import numpy as np
import torch

PI = math.pi
F64 = torch.float64


def _t(x):
    return torch.as_tensor(x, dtype=F64)


class _IeeeSqrt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.from_numpy(np.sqrt(x.detach().contiguous().numpy()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g / (2 * y)


def sqrt(x):
    """Correctly rounded float64 square root (see the module docstring), differentiable."""
    with np.errstate(invalid="ignore"):
        return _IeeeSqrt.apply(x)


def _grid(ray_field, boundary_field):
    """tf.meshgrid(ray_field, boundary_field) -> two (M, N) views."""
    r = _t(ray_field).reshape(1, -1)
    b = _t(boundary_field).reshape(-1, 1)
    shape = (b.shape[0], r.shape[1])
    return r.expand(shape), b.expand(shape)


# ---------------------------------------------------------------------------- lines

def raw_line_intersect(x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e, epsilion):
    """geometry.py:96-167.  Returns x, y, valid, u (first line), v (second line)."""
    x1s, y1s, x1e, y1e = _t(x1s), _t(y1s), _t(x1e), _t(y1e)
    x2s, y2s, x2e, y2e = _t(x2s), _t(y2s), _t(x2e), _t(y2e)
    x1 = x1e - x1s
    y1 = y1e - y1s
    x2 = x2e - x2s
    y2 = y2e - y2s
    denominator = x1 * y2 - y1 * x2

    valid = torch.abs(denominator) >= epsilion
    safe_value = torch.ones_like(denominator)
    safe_denominator = torch.where(valid, denominator, safe_value)
    safe_denominator = 1.0 / safe_denominator

    u = torch.where(
        valid, (x2 * (y1s - y2s) - y2 * (x1s - x2s)) * safe_denominator, safe_value
    )
    v = torch.where(
        valid, (y1 * (x2s - x1s) - x1 * (y2s - y1s)) * safe_denominator, safe_value
    )
    x = x1s + u * x1
    y = y1s + u * y1
    return x, y, valid, u, v


def line_intersect(x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e, epsilion):
    """geometry.py:27-78.  N first lines x M second lines -> (M, N) outputs."""
    x1s, x2s = _grid(x1s, x2s)
    y1s, y2s = _grid(y1s, y2s)
    x1e, x2e = _grid(x1e, x2e)
    y1e, y2e = _grid(y1e, y2e)
    return raw_line_intersect(x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e, epsilion)


# ------------------------------------------------------------------------ triangles

def raw_line_triangle_intersect(
    rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2, epsilion
):
    """geometry.py:275-320 (Cramer's rule with the reference's six-term sums).

    Returns x, y, z, valid, ray_u, trig_u, trig_v.
    """
    rx1, ry1, rz1, rx2, ry2, rz2 = (_t(v) for v in (rx1, ry1, rz1, rx2, ry2, rz2))
    xp, yp, zp, x1, y1, z1, x2, y2, z2 = (
        _t(v) for v in (xp, yp, zp, x1, y1, z1, x2, y2, z2)
    )
    a = rx1 - rx2
    b = x1 - xp
    c = x2 - xp
    d = ry1 - ry2
    f = y1 - yp
    g = y2 - yp
    h = rz1 - rz2
    k = z1 - zp
    l = z2 - zp

    q = rx1 - xp
    r = ry1 - yp
    s = rz1 - zp

    denominator = a * g * k + b * d * l + c * f * h - a * f * l - b * g * h - c * d * k
    ray_u_numerator = b * l * r + c * f * s + g * k * q - b * g * s - c * k * r - f * l * q
    trig_u_numerator = a * g * s + c * h * r + d * l * q - a * l * r - c * d * s - g * h * q
    trig_v_numerator = a * k * r + b * d * s + f * h * q - a * f * s - b * h * r - d * k * q

    valid = torch.abs(denominator) >= epsilion
    safe_value = torch.ones_like(denominator)
    safe_denominator = torch.where(valid, denominator, safe_value)
    ray_u = ray_u_numerator / safe_denominator
    trig_u = trig_u_numerator / safe_denominator
    trig_v = trig_v_numerator / safe_denominator

    x = rx1 - ray_u * a
    y = ry1 - ray_u * d
    z = rz1 - ray_u * h
    return x, y, z, valid, ray_u, trig_u, trig_v


def line_triangle_intersect(
    rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2, epsilion
):
    """geometry.py:191-251.  N rays x M triangles -> (M, N) outputs."""
    rx1_m, xp_m = _grid(rx1, xp)
    ry1_m, yp_m = _grid(ry1, yp)
    rz1_m, zp_m = _grid(rz1, zp)
    rx2_m, x1_m = _grid(rx2, x1)
    ry2_m, y1_m = _grid(ry2, y1)
    rz2_m, z1_m = _grid(rz2, z1)
    _, x2_m = _grid(rx2, x2)
    _, y2_m = _grid(ry2, y2)
    _, z2_m = _grid(rz2, z2)
    return raw_line_triangle_intersect(
        rx1_m, ry1_m, rz1_m, rx2_m, ry2_m, rz2_m,
        xp_m, yp_m, zp_m, x1_m, y1_m, z1_m, x2_m, y2_m, z2_m, epsilion,
    )


# -------------------------------------------------------------------------- circles

def raw_line_circle_intersect(xs, ys, xe, ye, xc, yc, r, epsilion):
    """geometry.py:420-547.  Returns (plus, minus) dicts with x, y, valid, u, v."""
    xs, ys, xe, ye, xc, yc, r = (_t(v) for v in (xs, ys, xe, ye, xc, yc, r))
    inverse_r = 1.0 / r
    xr = (xs - xc) * inverse_r
    yr = (ys - yc) * inverse_r
    xd = (xe - xs) * inverse_r
    yd = (ye - ys) * inverse_r

    a = xd * xd + yd * yd
    b = 2.0 * xr * xd + 2.0 * yr * yd
    c = xr * xr + yr * yr - 1.0
    rad = b * b - 4.0 * a * c

    # tangent snap
    rad = torch.where(torch.abs(rad) < epsilion, torch.zeros_like(rad), rad)

    safe_value = torch.ones_like(a)
    rad_less = rad < 0
    uminus_valid = uplus_valid = torch.logical_not(rad_less)
    safe_rad = sqrt(torch.where(rad_less, safe_value, rad))
    uminus = torch.where(rad_less, safe_value, (-b - safe_rad))
    uplus = torch.where(rad_less, safe_value, (-b + safe_rad))

    azero = torch.abs(a) < epsilion
    safe_denominator = 1.0 / torch.where(azero, safe_value, 2 * a)
    uminus_valid = torch.logical_and(uminus_valid, torch.logical_not(azero))
    uminus = torch.where(azero, safe_value, uminus * safe_denominator)
    uplus_valid = torch.logical_and(uplus_valid, torch.logical_not(azero))
    uplus = torch.where(azero, safe_value, uplus * safe_denominator)

    xminus = xs + (xe - xs) * uminus
    xplus = xs + (xe - xs) * uplus
    yminus = ys + (ye - ys) * uminus
    yplus = ys + (ye - ys) * uplus
    vminus = torch.atan2(yminus - yc, xminus - xc)
    vplus = torch.atan2(yplus - yc, xplus - xc)

    return (
        {"x": xplus, "y": yplus, "valid": uplus_valid, "u": uplus, "v": vplus},
        {"x": xminus, "y": yminus, "valid": uminus_valid, "u": uminus, "v": vminus},
    )


def line_circle_intersect(xs, ys, xe, ye, xc, yc, r, epsilion):
    """geometry.py:338-402.  N lines x M circles -> (M, N) outputs."""
    xs, xc = _grid(xs, xc)
    ys, yc = _grid(ys, yc)
    xe, _ = _grid(xe, r)
    ye, r = _grid(ye, r)
    return raw_line_circle_intersect(xs, ys, xe, ye, xc, yc, r, epsilion)


def angle_in_interval(angle, start, end):
    """geometry.py:766-802.  Inputs assumed in [-pi, pi]."""
    angle, start, end = _t(angle), _t(start), _t(end)
    reduced_angle = angle - start
    reduced_angle = torch.where(reduced_angle < 0.0, reduced_angle + 2 * PI, reduced_angle)
    reduced_end = end - start
    reduced_end = torch.where(reduced_end < 0.0, reduced_end + 2 * PI, reduced_end)
    return reduced_angle <= reduced_end


# ---------------------------------------------------------------------------- Snell

def snells_law_2D(x_start, y_start, x_end, y_end, norm, n_in, n_out, new_ray_length,
                  finite_tir_gradient=False):
    """geometry.py:565-653 (angle form).

    As in the reference, ``asin(theta2)`` is evaluated for every ray and is NaN in the
    unselected branch of a totally reflected one (geometry.py:640-646): the forward value is the
    reflect branch, but the GRADIENT of such a ray is NaN (0 * NaN in asin's derivative) and
    poisons every boundary entry the ray touched; optimizer.py:226-229 zeroes those entries.
    ``finite_tir_gradient=True`` (not the reference; the product's opt-in) feeds asin a safe
    value there, which leaves the reflect branch's own finite gradient."""
    x_start, y_start, x_end, y_end = (_t(v) for v in (x_start, y_start, x_end, y_end))
    norm, n_in, n_out = _t(norm), _t(n_in), _t(n_out)
    norm = torch.remainder(norm, 2 * PI)
    ray_angle = torch.atan2(y_start - y_end, x_start - x_end)
    ray_angle = torch.remainder(ray_angle, 2 * PI)
    theta1 = norm - ray_angle
    theta1 = torch.where(theta1 > PI, theta1 - (2 * PI), theta1)
    theta1 = torch.where(theta1 < -PI, theta1 + (2 * PI), theta1)

    internal_mask = torch.abs(theta1) >= PI / 2
    shape = theta1.shape
    one = torch.ones_like(theta1)
    zero = torch.zeros_like(theta1)

    n_in = n_in.expand(shape)
    n_in_is_safe = n_in != 0.0
    n_in_safe = torch.where(n_in_is_safe, n_in, one)
    n_out = n_out.expand(shape)
    n_out_is_safe = n_out != 0.0
    n_out_safe = torch.where(n_out_is_safe, n_out, one)

    n1 = torch.where(n_out_is_safe, n_in_safe / n_out_safe, zero)
    n2 = torch.where(n_in_is_safe, n_out_safe / n_in_safe, zero)
    n = torch.where(internal_mask, n1, n2)

    norm = torch.where(internal_mask, norm.expand(shape), (norm + PI).expand(shape))
    theta1 = torch.where(internal_mask, theta1 + PI, theta1)

    theta2 = n * torch.sin(theta1)
    ok = torch.logical_and(torch.abs(theta2) <= 1.0, n != 0.0)
    asin_arg = torch.where(ok, theta2, zero) if finite_tir_gradient else theta2
    new_angle = torch.where(ok, norm - torch.asin(asin_arg), norm + theta1 + PI)

    xs = x_end
    ys = y_end
    xe = xs + new_ray_length * torch.cos(new_angle)
    ye = ys + new_ray_length * torch.sin(new_angle)
    return xs, ys, xe, ye


def _l2_normalize(v, eps=1e-12):
    sq = torch.sum(v * v, dim=1, keepdim=True)
    return v * torch.rsqrt(torch.clamp(sq, min=eps))


def snells_law_3D(
    x_start, y_start, z_start, x_end, y_end, z_end, norm, n_in, n_out, new_ray_length
):
    """geometry.py:671-753 (vector form)."""
    x_start, y_start, z_start = _t(x_start), _t(y_start), _t(z_start)
    x_end, y_end, z_end = _t(x_end), _t(y_end), _t(z_end)
    norm, n_in, n_out = _t(norm), _t(n_in), _t(n_out)

    u = torch.stack([x_end - x_start, y_end - y_start, z_end - z_start], dim=1)
    u = _l2_normalize(u)
    n = _l2_normalize(norm)
    nu = torch.sum(n * u, dim=1, keepdim=True)

    internal_mask = nu > 0
    one = torch.ones_like(n_in)
    zero = torch.zeros_like(n_in)
    n_in_is_safe = n_in != 0.0
    n_in_safe = torch.where(n_in_is_safe, n_in, one)
    n_out_is_safe = n_out != 0.0
    n_out_safe = torch.where(n_out_is_safe, n_out, one)

    n1 = torch.where(n_out_is_safe, n_in_safe / n_out_safe, zero).reshape(-1, 1)
    n2 = torch.where(n_in_is_safe, n_out_safe / n_in_safe, zero).reshape(-1, 1)
    eta = torch.where(internal_mask, n1, n2)
    nu_eta = eta * nu

    radicand = 1 - eta * eta + nu_eta * nu_eta
    do_tir = radicand < 0
    safe_radicand = torch.where(do_tir, torch.ones_like(radicand), radicand)
    refract = (torch.sign(nu) * sqrt(safe_radicand) - nu_eta) * n + eta * u
    reflect = -2 * nu * n + u

    reflective_surface = (n_in == 0).reshape(-1, 1)
    do_reflect = torch.logical_or(do_tir, reflective_surface)
    new_vector = torch.where(do_reflect, reflect, refract)

    new_end = torch.stack([x_end, y_end, z_end], dim=1) + new_ray_length * new_vector
    return x_end, y_end, z_end, new_end[:, 0], new_end[:, 1], new_end[:, 2]
