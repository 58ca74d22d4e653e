"""
Angle / base-point clouds that feed the sources (subset of tfrt/distributions.py: the
distributions the hot-path configs use).  Host-side, O(N), runs once per ``update()``;
torch ops on the configured device.

Random distributions draw from a module-level ``torch.Generator`` (``seed(n)``) so runs and
ranks of a sharded job are reproducible.

Quaternion helpers follow the Hamilton convention (w, x, y, z).  The reference gets them from
the third-party ``tfquaternion`` package, which is not available here and which no reference
test pins: parity for rotated sources is unpinned.
"""
import math
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import config
from .update import RecursivelyUpdatable

PI = math.pi

_generator = None


def seed(value):
    """Seed the generator used by every Random* distribution."""
    global _generator
    _generator = torch.Generator(device="cpu")
    _generator.manual_seed(int(value))


def _uniform(n, low=0.0, high=1.0):
    global _generator
    if _generator is None:
        seed(1234)
    u = torch.rand(int(n), dtype=torch.float64, generator=_generator)
    return (low + (high - low) * u).to(config.get_device())


def _f64(x):
    return config.as_f64(x)


# ------------------------------------------------------------------------- quaternions

def quaternion_multiply(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def rotate_vector_by_quaternion(q, v):
    """v' = q v q*  for q = (w, x, y, z); v (..., 3)."""
    q = _f64(q)
    q = q / torch.linalg.norm(q)
    v = _f64(v)
    w = q[0]
    u = q[1:].expand(v.shape)
    t = 2.0 * torch.linalg.cross(u, v, dim=-1)
    return v + w * t + torch.linalg.cross(u, t, dim=-1)


def get_rotation_quaternion_from_u_to_v(u, v):
    """Shortest-arc quaternion rotating direction u onto direction v."""
    u = _f64(u)
    v = _f64(v)
    u = u / torch.linalg.norm(u)
    v = v / torch.linalg.norm(v)
    d = torch.dot(u, v)
    if float(d) < -1.0 + 1e-12:  # opposite: rotate pi about any axis orthogonal to u
        axis = torch.linalg.cross(u, _f64([1.0, 0.0, 0.0]))
        if float(torch.linalg.norm(axis)) < 1e-6:
            axis = torch.linalg.cross(u, _f64([0.0, 1.0, 0.0]))
        axis = axis / torch.linalg.norm(axis)
        return torch.cat([_f64([0.0]), axis])
    q = torch.cat([(1.0 + d).reshape(1), torch.linalg.cross(u, v)])
    return q / torch.linalg.norm(q)


def quaternion_from_euler(angles):
    """Rotations about x, then y, then z."""
    ax, ay, az = [float(a) for a in angles]
    qx = _f64([math.cos(ax / 2), math.sin(ax / 2), 0, 0])
    qy = _f64([math.cos(ay / 2), 0, math.sin(ay / 2), 0])
    qz = _f64([math.cos(az / 2), 0, 0, math.sin(az / 2)])
    return quaternion_multiply(qz, quaternion_multiply(qy, qx))


# ------------------------------------------------------------------------------ angles

class AngularDistributionBase(ABC):
    """2-D angular distributions: ``angles`` (radians) and ``ranks``
    (distributions.py:27-163)."""

    def __init__(self, min_angle, max_angle, sample_count, name=None):
        self.min_angle = min_angle
        self.max_angle = max_angle
        self.sample_count = sample_count
        self._name = name
        self.update()

    def angle_limit_validation(self, lower, upper):
        if not (lower <= float(self.min_angle) <= float(self.max_angle) <= upper):
            raise ValueError(
                f"AngularDistribution: need {lower} <= min_angle <= max_angle <= {upper}.")

    @staticmethod
    def _update_ranks(angles, min_angle, max_angle):
        return angles / max(abs(float(min_angle)), abs(float(max_angle)), 1e-300)

    @abstractmethod
    def update(self):
        raise NotImplementedError

    @property
    def name(self):
        return self._name

    @property
    def angles(self):
        return self._angles

    @property
    def ranks(self):
        return self._ranks


class ManualAngularDistribution(RecursivelyUpdatable):
    """Angles given directly (2-D scalars or 3-D direction vectors), distributions.py:166-237."""

    def __init__(self, angles, ranks=None, name=None, **kwargs):
        self._angles = _f64(angles)
        self._ranks = None if ranks is None else _f64(ranks)
        self._name = name
        super().__init__(**kwargs)

    def _update(self):
        pass

    def _generate_update_handles(self):
        return []

    angles = property(lambda self: self._angles)
    ranks = property(lambda self: self._ranks)
    name = property(lambda self: self._name)

    @angles.setter
    def angles(self, val):
        self._angles = _f64(val)


class StaticUniformAngularDistribution(AngularDistributionBase):
    """linspace(min_angle, max_angle, sample_count) (distributions.py:240-316)."""

    def update(self):
        self.angle_limit_validation(-PI, PI)
        self._angles = torch.linspace(float(self.min_angle), float(self.max_angle),
                                      int(self.sample_count), dtype=torch.float64,
                                      device=config.get_device())
        self._ranks = self._update_ranks(self._angles, self.min_angle, self.max_angle)


class RandomUniformAngularDistribution(AngularDistributionBase):
    def update(self):
        self.angle_limit_validation(-PI, PI)
        self._angles = _uniform(self.sample_count, float(self.min_angle), float(self.max_angle))
        self._ranks = self._update_ranks(self._angles, self.min_angle, self.max_angle)


# ------------------------------------------------------------------------- base points

class BasePointDistributionBase(RecursivelyUpdatable):
    """``points`` (N,2|3) and ``ranks`` (distributions.py:559-626)."""

    def __init__(self, name=None, **kwargs):
        self._name = name
        self._ranks = None
        super().__init__(**kwargs)

    def _generate_update_handles(self):
        return []

    def sample_count_validation(self):
        if int(self.sample_count) <= 0:
            raise ValueError("BasePointDistribution: sample_count must be > 0.")

    points = property(lambda self: self._points)
    ranks = property(lambda self: self._ranks)
    name = property(lambda self: self._name)


class ManualBasePointDistribution(BasePointDistributionBase):
    """Points given directly, or taken from a mesh's points (distributions.py:629-743)."""

    def __init__(self, dimension=None, points=None, ranks=None, from_mesh=False, **kwargs):
        if from_mesh:
            points = points.points
        self._points = _f64(points if points is not None else np.zeros((0, dimension or 2)))
        self._user_ranks = None if ranks is None else _f64(ranks)
        super().__init__(**kwargs)
        self._ranks = self._user_ranks

    def _update(self):
        self._ranks = self._user_ranks

    @BasePointDistributionBase.points.setter
    def points(self, val):
        self._points = _f64(val)


class BeamPointBase(BasePointDistributionBase):
    """Points on a line through the origin perpendicular to ``central_angle``
    (distributions.py:746-885)."""

    def __init__(self, beam_start, beam_end, sample_count, central_angle=0.0, **kwargs):
        self.beam_start = float(beam_start)
        self.beam_end = float(beam_end)
        self.sample_count = int(sample_count)
        self.central_angle = float(central_angle)
        super().__init__(**kwargs)

    def _update(self):
        if self.beam_start > self.beam_end:
            raise ValueError("BeamPointBase: beam_start must be < beam_end.")
        self.sample_count_validation()
        rank_scale = max(abs(self.beam_start), abs(self.beam_end))
        start_rank = self.beam_start / rank_scale
        end_rank = self.beam_end / rank_scale
        scale = self.beam_start / abs(start_rank)
        endpoint = _f64([scale * math.cos(self.central_angle - PI / 2.0),
                         scale * math.sin(self.central_angle - PI / 2.0)])
        self._ranks = self._update_ranks(start_rank, end_rank, self.sample_count)
        self._points = endpoint.reshape(1, 2) * self._ranks.reshape(-1, 1)

    @staticmethod
    @abstractmethod
    def _update_ranks(start_rank, end_rank, sample_count):
        raise NotImplementedError


class StaticUniformBeam(BeamPointBase):
    @staticmethod
    def _update_ranks(start_rank, end_rank, sample_count):
        return torch.linspace(start_rank, end_rank, sample_count, dtype=torch.float64,
                              device=config.get_device())


class RandomUniformBeam(BeamPointBase):
    @staticmethod
    def _update_ranks(start_rank, end_rank, sample_count):
        return _uniform(sample_count, start_rank, end_rank)


class AperaturePointBase(BasePointDistributionBase):
    """Points on the segment start_point -> end_point, rank 0..1
    (distributions.py:1019-1122)."""

    def __init__(self, start_point, end_point, sample_count, **kwargs):
        self.start_point = start_point
        self.end_point = end_point
        self.sample_count = int(sample_count)
        super().__init__(**kwargs)

    def _update(self):
        self.sample_count_validation()
        self._ranks = self._update_ranks(self.sample_count).reshape(-1, 1)
        self._points = self._start_point + self._ranks * (self._end_point - self._start_point)

    start_point = property(lambda self: self._start_point)
    end_point = property(lambda self: self._end_point)

    @start_point.setter
    def start_point(self, val):
        self._start_point = _f64(val).reshape(1, 2)

    @end_point.setter
    def end_point(self, val):
        self._end_point = _f64(val).reshape(1, 2)


class StaticUniformAperaturePoints(AperaturePointBase):
    @staticmethod
    def _update_ranks(sample_count):
        return torch.linspace(0.0, 1.0, sample_count, dtype=torch.float64,
                              device=config.get_device())


class RandomUniformAperaturePoints(AperaturePointBase):
    @staticmethod
    def _update_ranks(sample_count):
        return _uniform(sample_count)


class SquareBase(BasePointDistributionBase):
    """Points in an axis-aligned rectangle centred on the origin
    (distributions.py:1238-1358)."""

    def __init__(self, x_size, x_res, y_size=None, y_res=None, **kwargs):
        self.x_size = float(x_size)
        self.x_res = int(x_res)
        self.y_size = float(y_size if y_size is not None else x_size)
        self.y_res = int(y_res if y_res is not None else x_res)
        super().__init__(**kwargs)


class StaticUniformSquare(SquareBase):
    def _update(self):
        dev = config.get_device()
        x = torch.linspace(-1.0, 1.0, self.x_res, dtype=torch.float64, device=dev)
        y = torch.linspace(-1.0, 1.0, self.y_res, dtype=torch.float64, device=dev)
        gx, gy = torch.meshgrid(x, y, indexing="xy")
        self._ranks = torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)
        self._points = self._ranks * _f64([self.x_size / 2, self.y_size / 2])


class RandomUniformSquare(SquareBase):
    def _update(self):
        n = self.x_res * self.y_res
        self._ranks = torch.stack([_uniform(n, -1, 1), _uniform(n, -1, 1)], dim=1)
        self._points = self._ranks * _f64([self.x_size / 2, self.y_size / 2])


class ThetaMod:
    """theta_start/theta_end wedge support shared by circles and spheres
    (distributions.py:1396-1447)."""

    def _theta_mod(self, theta):
        if self.theta_start == 0 and self.theta_end == 2 * PI:
            return theta
        return torch.remainder(theta, self.theta_end - self.theta_start) + self.theta_start


class CircleBase(ThetaMod, RecursivelyUpdatable):
    """Golden-spiral disc of ``sample_count`` points (distributions.py:1450-1567)."""

    def __init__(self, sample_count, radius=1.0, theta_start=0, theta_end=2 * PI, **kwargs):
        if int(sample_count) <= 0:
            raise ValueError("CircleDistribution: sample_count must be > 0.")
        if float(radius) <= 0:
            raise ValueError("CircleDistribution: radius must be > 0.")
        self.sample_count = int(sample_count)
        self.radius = float(radius)
        self.theta_start = float(theta_start)
        self.theta_end = float(theta_end)
        RecursivelyUpdatable.__init__(self, **kwargs)

    def _generate_update_handles(self):
        return []

    def _finish(self):
        self._points = self.radius * torch.stack(
            [self._r * torch.cos(self._theta), self._r * torch.sin(self._theta)], dim=1)

    points = property(lambda self: self._points)

    @property
    def polar_points(self):
        return torch.stack([self.radius * self._r, torch.remainder(self._theta, 2 * PI)], dim=1)

    @property
    def ranks(self):
        return torch.stack([self._r * torch.cos(self._theta), self._r * torch.sin(self._theta)], dim=1)

    @property
    def polar_ranks(self):
        return torch.stack([self._r, torch.remainder(self._theta, 2 * PI)], dim=1)


class StaticUniformCircle(CircleBase):
    def _update(self):
        # deterministic: recompute only when a parameter changed (update() is called every
        # optimiser step by system.update(); the result would be identical)
        key = (self.sample_count, self.radius, self.theta_start, self.theta_end,
               str(config.get_device()))
        if getattr(self, "_memo_key", None) == key:
            self._points = self._memo_points
            return
        self._compute()
        self._memo_key, self._memo_points = key, self._points

    def _compute(self):
        idx = torch.arange(self.sample_count, dtype=torch.float64, device=config.get_device()) + .5
        self._r = torch.sqrt(idx / self.sample_count)
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * idx)
        self._finish()


class RandomUniformCircle(CircleBase):
    def _update(self):
        self._r = torch.sqrt(_uniform(self.sample_count))
        self._theta = self._theta_mod(2 * PI * _uniform(self.sample_count))
        self._finish()


class SphereBase(ThetaMod, RecursivelyUpdatable):
    """Direction vectors on a spherical cap about +x (distributions.py:1601-1723)."""

    def __init__(self, angular_size, sample_count, radius=1.0, theta_start=0, theta_end=2 * PI,
                 **kwargs):
        self.angular_size = float(angular_size)
        self.sample_count = int(sample_count)
        self.radius = float(radius)
        self.theta_start = float(theta_start)
        self.theta_end = float(theta_end)
        RecursivelyUpdatable.__init__(self, **kwargs)

    def _generate_update_handles(self):
        return []

    def _finish(self):
        self._points = self.radius * torch.stack([
            torch.cos(self._phi),
            torch.sin(self._phi) * torch.cos(self._theta),
            torch.sin(self._phi) * torch.sin(self._theta)], dim=1)

    points = property(lambda self: self._points)
    angles = property(lambda self: self._points)  # a sphere doubles as a 3-D angular distribution

    @property
    def ranks(self):
        return torch.stack([self._phi / self.angular_size, torch.remainder(self._theta, 2 * PI)], dim=1)


class StaticUniformSphere(SphereBase):
    def _update(self):
        dev = config.get_device()
        idx = torch.arange(self.sample_count, dtype=torch.float64, device=dev) + .5
        c = torch.linspace(1.0, math.cos(self.angular_size), self.sample_count,
                           dtype=torch.float64, device=dev)
        self._phi = torch.acos(c)
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * idx)
        self._finish()


class RandomUniformSphere(SphereBase):
    def _update(self):
        c = _uniform(self.sample_count, math.cos(self.angular_size), 1.0)
        self._phi = torch.acos(c)
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * _uniform(self.sample_count))
        self._finish()


class BasePointTransformation:
    """Lifts a 2-D base point distribution into 3-D (points assumed in the y-z plane) and
    optionally scales / rotates (quaternion) / translates it, by rewriting the base's points
    after each of its updates (distributions.py:2014-2120)."""

    def __init__(self, base, rotation=None, translation=None, scale=None):
        self._base = base
        self.rotation = rotation
        self.translation = translation
        self.scale = scale
        self._base.post_update_handles.append(self._apply_transformation)

    def _apply_transformation(self):
        pts = self._base._points
        key = (id(pts), pts._version, id(self._scale), id(self._rotation), id(self._translation))
        if getattr(self, "_memo_key", None) == key:
            self._base._points = self._memo_out
            return
        self._memo_in = pts  # keep the input alive so id() stays unique
        out = self._transform(pts)
        self._memo_key, self._memo_out = key, out
        self._base._points = out

    def _transform(self, pts):
        if pts.shape[1] == 2:
            pts = torch.cat([torch.zeros_like(pts[:, :1]), pts], dim=1)
        if self._scale is not None:
            pts = pts * self._scale.to(pts.device)
        if self._rotation is not None:
            pts = rotate_vector_by_quaternion(self._rotation.to(pts.device), pts)
        if self._translation is not None:
            pts = pts + self._translation.to(pts.device)
        return pts

    rotation = property(lambda self: self._rotation)
    translation = property(lambda self: self._translation)
    scale = property(lambda self: self._scale)

    @rotation.setter
    def rotation(self, val):
        if val is not None:
            val = _f64(val)
            if tuple(val.shape) != (4,):
                raise ValueError("BasePointTransformation: rotation must be a quaternion.")
        self._rotation = val

    @translation.setter
    def translation(self, val):
        self._translation = None if val is None else _f64(val)

    @scale.setter
    def scale(self, val):
        self._scale = None if val is None else _f64(val)
