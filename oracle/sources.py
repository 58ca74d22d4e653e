"""
numpy restatement of the ray-set assembly of tfrt/sources.py (TEST INFRASTRUCTURE -- see
oracle/__init__.py): the dense / undense domain logic of ``SourceBase._resize`` / ``make_vars`` /
``publish_extra_fields`` (sources.py:170-315) and the ``_internal_update`` of ``PointSource``
(:590-635, 2-D and 3-D without rotation), ``AngularSource`` (:820-866) and ``AperatureSource``
(:1045-1064).  3-D rotations are left out: the reference takes them from ``tfquaternion``,
which is not available here (parity unpinned for them); the configurations
of BASELINE.json use identity / translation-only placement.

Dense sources combine every domain with every other: ``tf.meshgrid(*ranges)`` with the default
'xy' indexing, flattened -- so with domains [d0, d1, d2, ...] in insertion order the FIRST TWO
axes are swapped (output shape (n1, n0, n2, ...)), which fixes the order of the generated rays.
"""
import numpy as np


def domain_sizes(internal_vars, extra_fields=None):
    """sources.py:186-230: per domain the one size that is not 1 (or 1); insertion-ordered."""
    sizes = {}
    items = list(internal_vars.values()) + list((extra_fields or {}).values())
    for domain, value in items:
        shape = np.shape(value)
        sizes.setdefault(domain, []).append(shape[0] if len(shape) else 1)
    out = {}
    for domain, ss in sizes.items():
        s = set(ss)
        if s == {1}:
            out[domain] = 1
        else:
            s -= {1}
            if len(s) != 1:
                raise ValueError("Source resize: found incompatible shapes in the same domain.")
            out[domain] = s.pop()
    return out


def dense_gathers(sizes):
    """sources.py:232-252: gather index per domain (all but "whole"), and the total ray count."""
    domains = [d for d in sizes if d != "whole"]
    grids = np.meshgrid(*[np.arange(sizes[d]) for d in domains])      # default indexing='xy'
    gathers = {d: g.reshape(-1) for d, g in zip(domains, grids)}
    whole = int(np.prod([sizes[d] for d in domains]))
    return gathers, whole


def undense_whole(sizes):
    """sources.py:254-272: every domain has the same size (or 1)."""
    whole = 1
    for s in sizes.values():
        if s == 1:
            continue
        if whole == 1:
            whole = s
        if whole != s:
            raise ValueError("Source resize: found incompatibly sized variables with an undense source.")
    return whole


def make_vars(internal_vars, dense, extra_fields=None):
    """sources.py:170-182 + 274-303.  Returns (expanded internal vars, expanded extra fields, N)."""
    sizes = domain_sizes(internal_vars, extra_fields)
    if dense:
        gathers, whole = dense_gathers(sizes)
    else:
        gathers, whole = {}, undense_whole(sizes)
    sizes = dict(sizes, whole=whole)
    out = {}
    for name, (domain, var) in internal_vars.items():
        var = np.asarray(var, dtype=np.float64)
        if dense:
            if var.ndim == 0:
                var = var.reshape(1)
            var = var[gathers[domain]]
        elif var.ndim < 2:
            var = np.broadcast_to(var, (whole,))
        out[name] = var
    extra = {}
    for field, (domain, value) in (extra_fields or {}).items():
        value = np.asarray(value)
        if value.ndim < 2:
            value = np.broadcast_to(value, (sizes[domain],))
        if domain != "whole" and dense:
            value = value[gathers[domain]]
        extra[field] = value
    return out, extra, whole


def aperature_source(start_points, end_points, wavelengths, dense, extra_fields=None):
    """AperatureSource._internal_update (sources.py:1045-1064)."""
    iv = {"start_points": ("start_point", start_points), "end_points": ("end_point", end_points)}
    if wavelengths is not None:
        iv["wavelengths"] = ("wavelength", wavelengths)
    v, extra, _ = make_vars(iv, dense, extra_fields)
    names = "xyz"[:np.shape(start_points)[1]]
    out = {}
    for k, a in enumerate(names):
        out[a + "_start"] = v["start_points"][:, k]
        out[a + "_end"] = v["end_points"][:, k]
    if wavelengths is not None:
        out["wavelength"] = v["wavelengths"]
    out.update(extra)
    return out


def point_source_2d(center, central_angle, angles, wavelengths, dense, start_on_center=True,
                    ray_length=1.0):
    """PointSource._internal_update, 2-D (sources.py:590-635)."""
    iv = {"angles": ("angle", angles)}
    if wavelengths is not None:
        iv["wavelengths"] = ("wavelength", wavelengths)
    v, _, _ = make_vars(iv, dense)
    ang = v["angles"] + central_angle
    xs = np.broadcast_to(center[0], ang.shape)
    ys = np.broadcast_to(center[1], ang.shape)
    xe = xs + ray_length * np.cos(ang)
    ye = ys + ray_length * np.sin(ang)
    if not start_on_center:
        xs, ys, xe, ye = xe, ye, xs, ys
    out = dict(x_start=xs, y_start=ys, x_end=xe, y_end=ye)
    if wavelengths is not None:
        out["wavelength"] = v["wavelengths"]
    return out


def angular_source_2d(center, central_angle, angles, base_points, wavelengths, dense,
                      start_on_base=True, ray_length=1.0):
    """AngularSource._internal_update, 2-D (sources.py:820-866): base points rotated by the
    central angle about the origin, then shifted to the centre."""
    iv = {"angles": ("angle", angles), "base_points": ("base_point", base_points)}
    if wavelengths is not None:
        iv["wavelengths"] = ("wavelength", wavelengths)
    v, _, _ = make_vars(iv, dense)
    ang = v["angles"] + central_angle
    c, s = np.cos(central_angle), np.sin(central_angle)
    bp = v["base_points"]
    rot = np.stack([c * bp[:, 0] - s * bp[:, 1], s * bp[:, 0] + c * bp[:, 1]], axis=1)
    start = np.asarray(center) + rot
    xs, ys = start[:, 0], start[:, 1]
    xe = xs + ray_length * np.cos(ang)
    ye = ys + ray_length * np.sin(ang)
    if not start_on_base:
        xs, ys, xe, ye = xe, ye, xs, ys
    out = dict(x_start=xs, y_start=ys, x_end=xe, y_end=ye)
    if wavelengths is not None:
        out["wavelength"] = v["wavelengths"]
    return out
