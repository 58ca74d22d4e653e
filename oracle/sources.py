"""
numpy restatement of the ray-set assembly of tfrt/sources.py (TEST INFRASTRUCTURE -- see
oracle/__init__.py): the dense / undense domain logic of ``SourceBase._resize`` / ``make_vars`` /
``publish_extra_fields`` (sources.py:170-315) and the ``_internal_update`` of ``PointSource``
(:590-635), ``AngularSource`` (:820-866) and ``AperatureSource`` (:1045-1064), and the static
point generators of tfrt/distributions.py (``StaticUniformSquare`` :1361-1372, ``StaticUniformSphere``
:1726-1747, ``StaticLambertianSphere`` :1778-1810).

3-D rotations (sources.py:428-458): the reference takes ``rotate_vector_by_quaternion`` and
``get_rotation_quaternion_from_u_to_v`` from the third-party package ``tfquaternion`` (unpinned:
not listed in the reference's requirements, absent from /root/reference and from this image).
They are restated here from their published definition -- Hamilton convention, q = (w, x, y, z),
v' = q (0, v) q*, and the shortest-arc quaternion normalise(|u||v| + u.v, u x v) -- and anchored on
what the reference's own call sites need of them: the quaternion of angle_type "vector" turns the
x axis into the central vector (sources.py:428-433), lengths and mutual angles of the rotated
directions and base points are kept, a "quaternion" source rotates by exactly the given
quaternion.  No reference test or fixture pins them: parity unpinned for 3-D source rotation.

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


# ---------------------------------------------------------------------------- quaternions

def quat_mul(a, b):
    """Hamilton product of quaternions (w, x, y, z); broadcasts over leading axes."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    aw, ax, ay, az = np.moveaxis(a, -1, 0)
    bw, bx, by, bz = np.moveaxis(b, -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def rotate_vector_by_quaternion(q, v):
    """tfq.rotate_vector_by_quaternion (sources.py:443, 458): v' = q (0, v) q* for the unit
    quaternion q / |q|; v (..., 3)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.sqrt((q * q).sum())
    v = np.asarray(v, dtype=np.float64)
    qv = np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)
    conj = q * np.array([1.0, -1.0, -1.0, -1.0])
    return quat_mul(quat_mul(q, qv), conj)[..., 1:]


def get_rotation_quaternion_from_u_to_v(u, v, eps=1e-6):
    """tfq.get_rotation_quaternion_from_u_to_v (sources.py:428): the shortest-arc rotation that
    turns direction u into direction v, normalise(|u||v| + u.v, u x v); for opposite directions a
    half turn about an axis perpendicular to u."""
    u, v = np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64)
    w = np.sqrt((u * u).sum() * (v * v).sum()) + (u * v).sum()
    if w < eps * np.sqrt((u * u).sum() * (v * v).sum()):
        axis = np.array([-u[1], u[0], 0.0]) if abs(u[0]) > abs(u[2]) else np.array([0.0, -u[2], u[1]])
        q = np.concatenate([[0.0], axis])
    else:
        q = np.concatenate([[w], np.cross(u, v)])
    return q / np.sqrt((q * q).sum())


def _central_quaternion(central_angle, angle_type):
    """RotationBase.central_angle setter, 3-D (sources.py:414-438)."""
    if angle_type == "vector":
        return get_rotation_quaternion_from_u_to_v(np.array([1.0, 0.0, 0.0]), central_angle)
    return np.asarray(central_angle, dtype=np.float64)


def point_source_3d(center, central_angle, angles, wavelengths, dense, start_on_center=True,
                    ray_length=1.0, angle_type="vector"):
    """PointSource._internal_update, 3-D (sources.py:590-635): the direction vectors of the angular
    distribution rotated by the central quaternion, rays from the centre."""
    iv = {"angles": ("angle", angles)}
    if wavelengths is not None:
        iv["wavelengths"] = ("wavelength", wavelengths)
    v, _, _ = make_vars(iv, dense)
    d = rotate_vector_by_quaternion(_central_quaternion(central_angle, angle_type), v["angles"])
    start = np.broadcast_to(np.asarray(center, dtype=np.float64), d.shape)
    end = start + ray_length * d
    if not start_on_center:
        start, end = end, start
    out = {a + "_start": start[:, k] for k, a in enumerate("xyz")}
    out.update({a + "_end": end[:, k] for k, a in enumerate("xyz")})
    if wavelengths is not None:
        out["wavelength"] = v["wavelengths"]
    return out


def angular_source_3d(center, central_angle, angles, base_points, wavelengths, dense,
                      start_on_base=True, ray_length=1.0, angle_type="vector"):
    """AngularSource._internal_update, 3-D (sources.py:820-866): base points (2-D ones lie in the
    y-z plane, sources.py:450-456) and directions rotated by the central quaternion, base points
    then shifted to the centre."""
    iv = {"angles": ("angle", angles), "base_points": ("base_point", base_points)}
    if wavelengths is not None:
        iv["wavelengths"] = ("wavelength", wavelengths)
    v, _, _ = make_vars(iv, dense)
    q = _central_quaternion(central_angle, angle_type)
    bp = v["base_points"]
    if bp.shape[1] == 2:
        bp = np.concatenate([np.zeros((bp.shape[0], 1)), bp], axis=1)
    start = np.asarray(center, dtype=np.float64) + rotate_vector_by_quaternion(q, bp)
    end = start + ray_length * rotate_vector_by_quaternion(q, v["angles"])
    if not start_on_base:
        start, end = end, start
    out = {a + "_start": start[:, k] for k, a in enumerate("xyz")}
    out.update({a + "_end": end[:, k] for k, a in enumerate("xyz")})
    if wavelengths is not None:
        out["wavelength"] = v["wavelengths"]
    return out


# ----------------------------------------------------------------------- point generators

def static_uniform_square(x_size, x_res, y_size=None, y_res=None):
    """StaticUniformSquare._make_points (distributions.py:1361-1372): meshgrid ('xy') of two
    linspaces, flattened row by row."""
    y_size = x_size if y_size is None else y_size
    y_res = x_res if y_res is None else y_res
    x, y = np.meshgrid(np.linspace(-x_size, x_size, x_res), np.linspace(-y_size, y_size, y_res))
    return np.stack([x.reshape(-1), y.reshape(-1)], axis=1)


def _sphere_points(cos_phi, count, radius, theta_start, theta_end):
    theta = np.pi * (1 + 5 ** 0.5) * (np.arange(count, dtype=np.float64) + 0.5)
    if not (theta_start == 0 and theta_end == 2 * np.pi):         # ThetaMod, distributions.py:1431-1447
        theta = np.mod(theta, theta_end - theta_start) + theta_start
    phi = np.arccos(cos_phi)
    return radius * np.stack([np.cos(phi), np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta)],
                             axis=1)


def static_uniform_sphere(count, angular_size, radius=1.0, theta_start=0.0, theta_end=2 * np.pi):
    """StaticUniformSphere._update (distributions.py:1726-1747): cos(phi) uniform in
    [cos(angular_size), 1], golden-angle azimuths."""
    return _sphere_points(np.linspace(1.0, np.cos(angular_size), count), count, radius,
                          theta_start, theta_end)


def static_lambertian_sphere(count, angular_size, radius=1.0, theta_start=0.0, theta_end=2 * np.pi):
    """StaticLambertianSphere._update (distributions.py:1794-1810): cos^2(phi) uniform."""
    return _sphere_points(np.sqrt(np.linspace(1.0, np.cos(angular_size) ** 2, count)), count,
                          radius, theta_start, theta_end)
