"""
Angle / base-point clouds that feed the sources (tfrt/distributions.py).  Host-side, O(N),
runs once per ``update()``; torch ops on the configured device, numpy/scipy for the
density-inversion helpers the reference also does in numpy.

Random distributions draw from a per-device ``torch.Generator`` seeded by ``seed(n)`` so runs
and ranks of a sharded job are reproducible.

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

_seed = 1234
_generators = {}


def seed(value):
    """Seed the generator used by every Random* distribution."""
    global _seed
    _seed = int(value)
    _generators.clear()
    _streams[0] = 0       # (distributions made after this call draw the same streams again)


def _generator_for(dev):
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    gen = _generators.get(dev)
    if gen is None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(_seed)
        _generators[dev] = gen
    return gen, dev


def _uniform(n, low=0.0, high=1.0):
    # drawn on the device the rays live on (Philox stream of a per-device generator seeded by
    # seed(); a host draw + copy costs ~0.2 ms per 100k samples, four of them per random
    # aperture source and step): the same seed gives the same rays on every GPU of a sharded job
    gen, dev = _generator_for(config.get_device())
    u = torch.rand(int(n), dtype=torch.float64, generator=gen, device=dev)
    return low + (high - low) * u


def _f64(x):
    return config.as_f64(x)


# ------------------------------------------------------- random distributions on the device
#
# On a HIP device the Random* point distributions do not draw with stock tensor ops: a
# distribution is a small *program* (csrc/tfrt_source.hip, tfrt_points_program) whose sample i at
# update e is a pure function of (seed, stream, e, i) -- a counter-based generator.  ``update()``
# only steps the distribution's epoch counter (on the device); ``points`` / ``ranks`` / ...
# are made by one kernel when somebody asks, and a source made of such distributions writes its
# rays straight into persistent buffers (sources.DeviceRaySet), in any order.  The numbers differ
# from the torch generator's stream; the distributions are the same (tests/test_gpu_source_programs.py).

_device_random = True
_streams = [0]


def set_device_random(on):
    """False: the Random* distributions draw with torch ops again (new tensors per update)."""
    global _device_random
    _device_random = bool(on)


class _Drawn:
    """Attribute that is a stored tensor in the torch path and a kernel's output, made on first
    use after every update, in the device path."""

    def __init__(self, what):
        self.what = what

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        if obj.__dict__.get("_device_active"):
            return obj._draw(self.what)
        try:
            return obj.__dict__[self.name]
        except KeyError:
            raise AttributeError(self.name) from None

    def __set__(self, obj, val):
        obj.__dict__[self.name] = val


class _DeviceRandom:
    """Mixin of the Random* base point / direction distributions (see above)."""

    _kind = None          # _lib.PTS_*

    def _device_mode(self):
        # (a program bakes the transformation's scale / rotation / translation and the
        # distribution's parameters in as numbers: when one of them requires grad the torch
        # formulas run instead, so that autograd still reaches it)
        tr = self.__dict__.get("_transformations", ())
        baked = [getattr(t, name, None) for t in tr for name in ("scale", "rotation", "translation")]
        baked.extend(self.__dict__.get(name) for name in
                     ("radius", "theta_start", "theta_end", "angular_size", "x_size", "y_size"))
        if any(isinstance(t, torch.Tensor) and t.requires_grad for t in baked):
            return False
        return (_device_random and config.get_device().type == "cuda" and len(tr) <= 1)

    def _device_update(self):
        from . import _lib
        dev = config.get_device()
        ep = self.__dict__.get("_epoch_dev")
        if ep is None or ep.device != dev:
            self._epoch_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            _streams[0] += 1
            self._stream_id = _streams[0]
        self._device_active = True
        # (the device counter is stepped when somebody draws -- a source does it for all of its
        # distributions in one launch)
        self._epoch_pending = self.__dict__.get("_epoch_pending", 0) + 1
        self.epoch = self.__dict__.get("epoch", 0) + 1
        self._drawn = {}
        # (a transformation takes effect with the update after it was attached, like the
        # reference's post-update handle)
        self._active_transformations = list(self.__dict__.get("_transformations", ()))
        assert self._kind in (_lib.PTS_CIRCLE, _lib.PTS_SQUARE, _lib.PTS_SPHERE_UNIFORM,
                              _lib.PTS_SPHERE_LAMBERT)

    def _leave_device_mode(self):
        self.__dict__["_device_active"] = False

    def pending_epochs(self):
        """[(device counter, steps it is behind)] -- consumed by whoever flushes."""
        n, self._epoch_pending = self.__dict__.get("_epoch_pending", 0), 0
        return [(self._epoch_dev, n)] if n else []

    def flush_epoch(self):
        from . import ops
        for counter, n in self.pending_epochs():
            for _ in range(n):
                ops.epoch_advance([counter])

    def _program_parameters(self):
        raise NotImplementedError

    def program(self):
        """The distribution as a tfrt_points_program (with its BasePointTransformation).  Built
        once per set of parameters (reading the transformation's tensors back costs a host sync,
        which a captured launch sequence must not contain)."""
        from . import _lib
        tr = self.__dict__.get("_active_transformations", [])
        tensors = []
        if tr:
            tensors = [tr[0].scale, tr[0].rotation, tr[0].translation]
        key = (self._kind, self._stream_id, self._sample_total(),
               tuple(float(v) for v in self._program_parameters()), _seed,
               self._epoch_dev.data_ptr(),
               tuple((id(t), getattr(t, "_version", None)) for t in tensors))
        cached = self.__dict__.get("_program_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        pg = _lib.PointsProgram()
        pg.kind = self._kind
        pg.stream = self._stream_id
        pg.count = self._sample_total()
        pg.table = None
        for k, v in enumerate(self._program_parameters()):
            pg.p[k] = float(v)
        pg.has_scale = pg.has_quat = pg.has_shift = 0
        if tr:
            t = tr[0]
            if t.scale is not None:
                sc = t.scale.detach().cpu().reshape(-1).tolist()
                sc = sc * 3 if len(sc) == 1 else sc
                pg.has_scale = 1
                for k in range(3):
                    pg.scale[k] = sc[k]
            if t.rotation is not None:
                q = t.rotation.detach().cpu().double()
                q = (q / torch.linalg.norm(q)).tolist()
                pg.has_quat = 1
                for k in range(4):
                    pg.quat[k] = q[k]
            if t.translation is not None:
                sh = t.translation.detach().cpu().reshape(-1).tolist()
                pg.has_shift = 1
                for k in range(3):
                    pg.shift[k] = sh[k]
        pg.seed = _seed & 0xFFFFFFFFFFFFFFFF
        pg.epoch = self._epoch_dev.data_ptr()
        self._program_cache = (key, pg, tensors)
        return pg

    def _transformed(self):
        return bool(self.__dict__.get("_active_transformations"))

    def _draw(self, what):
        from . import ops, _lib
        d = self._drawn
        if what not in d:
            self.flush_epoch()
            cols = 3 if (self._transformed() or self._kind in (_lib.PTS_SPHERE_UNIFORM,
                                                               _lib.PTS_SPHERE_LAMBERT)) else 2
            pts, a0, a1 = ops.points_generate(self.program(), self._sample_total(), columns=cols,
                                              want_aux=True, device=self._epoch_dev.device)
            d["points"], d["aux0"], d["aux1"] = pts, a0, a1
        return d[what]


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


def get_rotation_quaternion_from_u_to_v(u, v, eps=1e-6):
    """Shortest-arc quaternion rotating direction u onto direction v: normalise(|u||v| + u.v,
    u x v) -- tfquaternion's definition, as far as it is published (the package is not available
    here; oracle/sources.py states the same).  Opposite directions: a half turn about
    (-u_y, u_x, 0), or (0, -u_z, u_y) when |u_x| <= |u_z|."""
    u = _f64(u)
    v = _f64(v)
    scale = torch.sqrt(torch.dot(u, u) * torch.dot(v, v))
    w = scale + torch.dot(u, v)
    if float(w) < eps * float(scale):
        zero = torch.zeros((), dtype=u.dtype, device=u.device)
        if abs(float(u[0])) > abs(float(u[2])):
            axis = torch.stack([-u[1], u[0], zero])
        else:
            axis = torch.stack([zero, -u[2], u[1]])
        q = torch.cat([zero.reshape(1), axis])
    else:
        q = torch.cat([w.reshape(1), torch.linalg.cross(u, v)])
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


class StaticLambertianAngularDistribution(AngularDistributionBase):
    """Cosine-weighted fan: the rank is sin(angle), spaced evenly between sin(min_angle) and
    sin(max_angle) (distributions.py:394-470).  Limits must lie in [-pi/2, pi/2]."""

    def update(self):
        self.angle_limit_validation(-PI / 2.0, PI / 2.0)
        self._ranks = torch.linspace(math.sin(float(self.min_angle)),
                                     math.sin(float(self.max_angle)), int(self.sample_count),
                                     dtype=torch.float64, device=config.get_device())
        self._angles = torch.asin(self._ranks)


class RandomLambertianAngularDistribution(AngularDistributionBase):
    """As above with the ranks drawn uniformly (distributions.py:473-556); re-sampled at
    every update."""

    def update(self):
        self.angle_limit_validation(-PI / 2.0, PI / 2.0)
        self._ranks = _uniform(self.sample_count, math.sin(float(self.min_angle)),
                               math.sin(float(self.max_angle)))
        self._angles = torch.asin(self._ranks)


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
    """x_size / y_size are centre-to-edge distances; the ranks are the points divided by the longer
    of the two (distributions.py:1352-1372)."""

    def _update(self):
        dev = config.get_device()
        x = torch.linspace(-self.x_size, self.x_size, self.x_res, dtype=torch.float64, device=dev)
        y = torch.linspace(-self.y_size, self.y_size, self.y_res, dtype=torch.float64, device=dev)
        gx, gy = torch.meshgrid(x, y, indexing="xy")
        self._points = torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)
        self._ranks = self._points / max(self.x_size, self.y_size)


class RandomUniformSquare(_DeviceRandom, SquareBase):
    _kind = 2     # _lib.PTS_SQUARE
    _points = _Drawn("points")

    def _sample_total(self):
        return self.x_res * self.y_res

    def _program_parameters(self):
        return (self.x_size, 0.0, 0.0, self.y_size)

    def _update(self):
        if self._device_mode():
            self._device_update()
            return
        self._leave_device_mode()
        n = self.x_res * self.y_res
        self._points = torch.stack([_uniform(n, -self.x_size, self.x_size),
                                    _uniform(n, -self.y_size, self.y_size)], dim=1)
        self._ranks = self._points / max(self.x_size, self.y_size)

    @property
    def ranks(self):
        if self.__dict__.get("_device_active"):
            return torch.stack([self._draw("aux0"), self._draw("aux1")], dim=1) / max(self.x_size, self.y_size)
        return self._ranks


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


class RandomUniformCircle(_DeviceRandom, CircleBase):
    _kind = 1     # _lib.PTS_CIRCLE
    _points = _Drawn("points")
    _r = _Drawn("aux0")
    _theta = _Drawn("aux1")

    def _sample_total(self):
        return self.sample_count

    def _program_parameters(self):
        return (self.radius, self.theta_start, self.theta_end, 0.0)

    def _update(self):
        if self._device_mode():
            self._device_update()
            return
        self._leave_device_mode()
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
        return torch.stack([self._phi, torch.remainder(self._theta, 2 * PI)], dim=1)


class StaticUniformSphere(SphereBase):
    def _update(self):
        dev = config.get_device()
        idx = torch.arange(self.sample_count, dtype=torch.float64, device=dev) + .5
        c = torch.linspace(1.0, math.cos(self.angular_size), self.sample_count,
                           dtype=torch.float64, device=dev)
        self._phi = torch.acos(c)
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * idx)
        self._finish()


class RandomUniformSphere(_DeviceRandom, SphereBase):
    _kind = 3     # _lib.PTS_SPHERE_UNIFORM
    _points = _Drawn("points")
    _phi = _Drawn("aux0")
    _theta = _Drawn("aux1")

    def _sample_total(self):
        return self.sample_count

    def _program_parameters(self):
        return (self.radius, self.theta_start, self.theta_end, math.cos(self.angular_size))

    def _update(self):
        if self._device_mode():
            self._device_update()
            return
        self._leave_device_mode()
        c = _uniform(self.sample_count, math.cos(self.angular_size), 1.0)
        self._phi = torch.acos(c)
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * _uniform(self.sample_count))
        self._finish()


class StaticLambertianSphere(SphereBase):
    """Golden-spiral cap whose polar density follows Lambert's cosine law: cos^2(phi) is
    spaced evenly from 1 to cos^2(angular_size) (distributions.py:1778-1811)."""

    def _update(self):
        dev = config.get_device()
        idx = torch.arange(self.sample_count, dtype=torch.float64, device=dev) + .5
        c2 = torch.linspace(1.0, math.cos(self.angular_size) ** 2, self.sample_count,
                            dtype=torch.float64, device=dev)
        self._phi = torch.acos(torch.sqrt(c2))
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * idx)
        self._finish()


class RandomLambertianSphere(_DeviceRandom, SphereBase):
    """distributions.py:1814-1850."""
    _kind = 4     # _lib.PTS_SPHERE_LAMBERT
    _points = _Drawn("points")
    _phi = _Drawn("aux0")
    _theta = _Drawn("aux1")

    def _sample_total(self):
        return self.sample_count

    def _program_parameters(self):
        return (self.radius, self.theta_start, self.theta_end, math.cos(self.angular_size) ** 2)

    def _update(self):
        if self._device_mode():
            self._device_update()
            return
        self._leave_device_mode()
        c2 = _uniform(self.sample_count, math.cos(self.angular_size) ** 2, 1.0)
        self._phi = torch.acos(torch.sqrt(c2))
        self._theta = self._theta_mod(PI * (1 + 5 ** 0.5) * _uniform(self.sample_count))
        self._finish()


class SquareRankLambertianSphere(RecursivelyUpdatable):
    """Lambertian direction cloud whose ranks fill the square [-1,1]^2 uniformly
    (distributions.py:1853-2011): uniform square -> (via an ArbitraryDistribution of a disc of
    radius sin(angular_cutoff)) uniform disc -> lifted onto the unit sphere, which makes the
    polar density cosine-weighted."""

    def __init__(self, sample_count, angular_cutoff=PI / 2.0, sampling_resolution=256, **kwargs):
        self.sampling_resolution = sampling_resolution
        self.angular_cutoff = angular_cutoff
        self.sample_count = sample_count
        RecursivelyUpdatable.__init__(self, **kwargs)

    sample_count = property(lambda self: self._sample_count)
    angular_cutoff = property(lambda self: self._angular_cutoff)
    sampling_resolution = property(lambda self: self._sampling_resolution)
    points = property(lambda self: self._points)
    angles = property(lambda self: self._points)
    ranks = property(lambda self: self._ranks)

    @sample_count.setter
    def sample_count(self, val):
        if val < 0:
            raise ValueError("SquareRankLambertianSphere: Sample count must be > 0.")
        self._sample_count = int(val)

    @sampling_resolution.setter
    def sampling_resolution(self, val):
        if val < 0:
            raise ValueError("SquareRankLambertianSphere: Sample count must be > 0.")
        self._sampling_resolution = int(val)

    @angular_cutoff.setter
    def angular_cutoff(self, val):
        if val > PI / 2.0 or val < 0:
            raise ValueError(
                "SquareRankLambertianSphere: angular cutoff must be between zero and PI/2.")
        self._angular_cutoff = float(val)
        limit = math.sin(self._angular_cutoff)

        def density(x, y):
            return (np.sqrt(x * x + y * y) < limit).astype(np.float64) + 1e-10

        res = self._sampling_resolution
        self._circle_maker = ArbitraryDistribution(density, ((-1.0, 1.0, res), (-1.0, 1.0, res)))

    def _update(self):
        n = self._sample_count
        ranks = np.stack([_uniform(n, -1.0, 1.0).cpu().numpy(),
                          _uniform(n, -1.0, 1.0).cpu().numpy()], axis=1)
        cx, cy = self._circle_maker(ranks[:, 0], ranks[:, 1])
        cx, cy = _f64(cx), _f64(cy)
        self._ranks = _f64(ranks)
        theta = torch.atan2(cy, cx)
        rad2 = cx * cx + cy * cy
        phi = torch.atan2(torch.sqrt(rad2), torch.sqrt(torch.clamp(1.0 - rad2, min=0.0)))
        self._phi = phi
        self._points = torch.stack([torch.cos(phi), torch.sin(phi) * torch.cos(theta),
                                    torch.sin(phi) * torch.sin(theta)], dim=1)

    def _generate_update_handles(self):
        return []


# --------------------------------------------------------------- arbitrary densities (host)

def _as_np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _read_grey_image(filename):
    """Grey levels (ITU-R 601 luma, float) of an image file; the reference uses
    ``imageio.imread(..., as_gray=True)`` (distributions.py:2191, 2930), which is not
    installed here, Pillow is."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading an image file needs Pillow (PIL)") from e
    with Image.open(filename) as im:
        return np.array(im.convert("F"), dtype=np.float64)


def _cumulative_tables(density):
    """Zero-started cumulative sums used by every density -> quantile mapping below
    (distributions.py:2230-2236): cumulate along axis 0 for each column, and cumulate the
    column totals along axis 1."""
    padded = np.pad(density, ((1, 0), (1, 0)), mode="constant", constant_values=0)
    per_column = np.cumsum(padded, axis=0)
    totals = np.cumsum(per_column[-1])
    return totals, per_column[:, 1:]


class ArbitraryDistribution:
    """Maps a uniform point cloud on a rectangle onto one that follows a 2-D density
    (distributions.py:2123-2280): the density (callable sampled on a grid, 2-D array, or grey
    image file) is cumulated, the cumulative curves are inverted by linear interpolation into
    an x quantile function and one y quantile function per x cell.  Host-side numpy/scipy,
    as in the reference."""

    def __init__(self, density_function, evaluation_limits):
        from scipy.interpolate import interp1d
        if type(density_function) is str:
            self._x_min, self._x_max = evaluation_limits[0]
            self._y_min, self._y_max = evaluation_limits[1]
            density = _read_grey_image(density_function)
            self._x_count, self._y_count = density.shape
        elif callable(density_function):
            self._x_min, self._x_max, self._x_count = evaluation_limits[0]
            self._y_min, self._y_max, self._y_count = evaluation_limits[1]
            gx, gy = np.meshgrid(np.linspace(self._x_min, self._x_max, self._x_count),
                                 np.linspace(self._y_min, self._y_max, self._y_count))
            density = np.asarray(density_function(gx, gy), dtype=np.float64)
        else:
            density = np.array(_as_np(density_function), dtype=np.float64)
            if density.ndim != 2:
                raise ValueError("PointCloudSampler: density function must be 2D.")
            self._x_min, self._x_max = evaluation_limits[0]
            self._y_min, self._y_max = evaluation_limits[1]
            self._x_count, self._y_count = density.shape
        if np.any(density < 0):
            raise ValueError("PointCloudSampler: density function must be non-negative on the "
                             "whole evaluation grid.")
        self.density_function = density
        totals, per_column = _cumulative_tables(density)
        totals = self._rescale(totals, self._x_min, self._x_max)
        columns = [self._rescale(per_column[:, i], self._y_min, self._y_max)
                   for i in range(self._x_count)]
        knots_x = np.linspace(self._x_min, self._x_max, self._x_count + 1)
        knots_y = np.linspace(self._y_min, self._y_max, self._y_count + 1)
        self._x_quantile = interp1d(totals, knots_x)
        self._y_quantiles = [interp1d(c, knots_y) for c in columns]

    @staticmethod
    def _rescale(n, n_min, n_max):
        top = np.amax(n)
        if top <= 0:
            raise ValueError(
                "PointCloudSampler: Discovered a slice where the density was zero, which causes "
                "problems, because the quantile function would have to have infinite slope.  "
                "Either restrict the evaluation range, or add a very small constant to the "
                "density function.")
        return n * (n_max - n_min) / top + n_min

    def __call__(self, x, y):
        x, y = np.asarray(_as_np(x), dtype=np.float64), np.asarray(_as_np(y), dtype=np.float64)
        x_out = self._x_quantile(x)
        cell = np.floor((x_out - self._x_min) * self._x_count
                        / (self._x_max - self._x_min)).astype(int)
        y_out = np.zeros_like(y)
        for i in range(self._y_count):
            pick = cell == i
            if pick.any():
                y_out[pick] = self._y_quantiles[i](y[pick])
        return x_out, y_out


def flatten_distribution(x, y, evaluation_limits):
    """Inverse of ArbitraryDistribution for an empirical cloud: histogram (x, y), build the
    cumulative curves and push the cloud through them so it becomes uniform on [0,1]^2
    (distributions.py:2283-2369, including its transposed-histogram convention)."""
    from scipy.interpolate import interp1d
    x_min, x_max, x_res = evaluation_limits[0]
    y_min, y_max, y_res = evaluation_limits[1]
    x = np.clip(_as_np(x), x_min, x_max)
    y = np.clip(_as_np(y), y_min, y_max)
    hist, _, _ = np.histogram2d(x, y, bins=(x_res, y_res),
                                range=((x_min, x_max), (y_min, y_max)))
    totals, per_column = _cumulative_tables(hist.T)
    totals = totals / np.amax(totals)
    columns = [per_column[:, i] / np.amax(per_column[:, i]) for i in range(x_res)]
    knots_x = np.linspace(x_min, x_max, x_res + 1)
    knots_y = np.linspace(y_min, y_max, y_res + 1)
    x_cdf = interp1d(knots_x, totals)
    y_cdfs = [interp1d(knots_y, c) for c in columns]
    x_out = x_cdf(x)
    cell = np.floor((x_out - x_min) * x_res / (x_max - x_min)).astype(int)
    y_out = np.zeros_like(y)
    for i in range(y_res):
        pick = cell == i
        if pick.any():
            y_out[pick] = y_cdfs[i](y[pick])
    return x_out, y_out


class CumulativeDensityFunction:
    """Forward (uniform -> density) and inverse (density -> uniform) maps of a 2-D histogram
    that can be accumulated over several batches (distributions.py:2372-2632).  ``cdf`` maps y
    first, then x with the curve of the y cell it landed in."""

    def __init__(self, eval_limits, density=None, direction="both"):
        self.x_res = 10
        self.y_res = 10
        self._y_cdf = self._x_cdfs = self._y_icdf = self._x_icdfs = None
        self.x_min, self.x_max = eval_limits[0]
        self.y_min, self.y_max = eval_limits[1]
        self._density = None
        if density is not None:
            self.compute(density, direction)

    def accumulate_density(self, density):
        d = np.array(_as_np(density), dtype=np.float32)
        if self._density is None:
            self._density = d
            self.x_res, self.y_res = d.shape
        else:
            self._density += d

    def clear_density(self):
        self._density = None

    def compute(self, density=None, direction="both", epsilon=1e-10):
        from scipy.interpolate import interp1d
        if density is not None:
            self.clear_density()
            self.accumulate_density(density)
        if direction not in {"forward", "inverse", "both"}:
            raise ValueError("CumulativeDensityFunction: direction must be one of {'forward', "
                             "'backward', 'both'}")
        if self._density is None:
            raise RuntimeError("CumulativeDensityFunction: cannot call compute before "
                               "accumulating data.")
        y_sum, x_sums = _cumulative_tables(self._density + epsilon)
        y_sum = y_sum / y_sum[-1]
        x_sums = x_sums / x_sums[-1:]
        knots_x = np.linspace(self.x_min, self.x_max, self.x_res + 1)
        knots_y = np.linspace(self.y_min, self.y_max, self.y_res + 1)
        if direction in {"forward", "both"}:
            self._y_cdf = interp1d(y_sum, knots_y)
            self._x_cdfs = [interp1d(x_sums[:, i], knots_x) for i in range(self.y_res)]
        else:
            self._y_cdf = self._x_cdfs = None
        if direction in {"inverse", "both"}:
            self._y_icdf = interp1d(knots_y, y_sum)
            self._x_icdfs = [interp1d(knots_x, x_sums[:, i]) for i in range(self.y_res)]
        else:
            self._y_icdf = self._x_icdfs = None

    def _apply(self, points, y_map, x_maps, cell_of):
        pts = _as_np(points)
        x, y = np.asarray(pts[:, 0], dtype=np.float64), np.asarray(pts[:, 1], dtype=np.float64)
        y_out = y_map(y)
        cell = np.floor(cell_of(y_out)).astype(int)
        x_out = np.zeros_like(x)
        for i in range(self.y_res):
            pick = cell == i
            if pick.any():
                x_out[pick] = x_maps[i](x[pick])
        return np.column_stack((x_out.astype(pts.dtype), y_out.astype(pts.dtype)))

    def cdf(self, points):
        if self._y_cdf is None:
            raise RuntimeError("CumulativeDensityFunction: Must call compute() with the correct "
                               "direction before evaluation.")
        return self._apply(points, self._y_cdf, self._x_cdfs,
                           lambda yo: (yo - self.y_min) * self.y_res / (self.y_max - self.y_min))

    def icdf(self, points):
        if self._y_icdf is None:
            raise RuntimeError("CumulativeDensityFunction: Must call compute() with the correct "
                               "direction before evaluation.")
        return self._apply(points, self._y_icdf, self._x_icdfs, lambda yo: yo * (self.y_res - 1))

    def __call__(self, points):
        return self.cdf(points)


class ArbitraryBasePoints(BasePointDistributionBase):
    """Base points following an ArbitraryDistribution, with ranks from a second one evaluated
    at the same uniform seeds (distributions.py:2635-2798); ``enforce_etendue`` rescales the
    ranks so their mean distance from ``origin`` equals the points'."""

    def __init__(self, base_point_distribution, sample_count, rank_distribution=None,
                 auto_reroll=True, conserve_etendue=True, etendue_origin=(0, 0), **kwargs):
        self.sample_count = sample_count
        self.base_point_distribution = base_point_distribution
        self.rank_distribution = rank_distribution
        self.auto_reroll = auto_reroll
        self.rank_scale_factor = 1
        super().__init__(**kwargs)
        if conserve_etendue:
            self.enforce_etendue(etendue_origin)

    sample_count = property(lambda self: self._sample_count)

    @sample_count.setter
    def sample_count(self, val):
        if int(val) != val or val <= 0:
            raise ValueError("AribitraryBasePoints: sample_count must be an integer > 0.")
        self._sample_count = int(val)

    def reroll(self):
        d = self.base_point_distribution
        self._base_x = _uniform(self._sample_count, d._x_min, d._x_max).cpu().numpy()
        self._base_y = _uniform(self._sample_count, d._y_min, d._y_max).cpu().numpy()

    def _update(self):
        if self.auto_reroll or self._ranks is None:
            self.reroll()
        self._points = _f64(np.stack(self.base_point_distribution(self._base_x, self._base_y), 1))
        if self.rank_distribution is not None:
            self._ranks = self.rank_scale_factor * _f64(
                np.stack(self.rank_distribution(self._base_x, self._base_y), 1))
        else:
            self._ranks = None

    def enforce_etendue(self, origin=(0, 0)):
        if self._ranks is not None:
            o = _f64(origin)
            base = torch.linalg.norm(self._points - o, dim=1).mean()
            ranks = torch.linalg.norm(self._ranks - o, dim=1).mean()
            self.rank_scale_factor = float(base / ranks)
            self._ranks = self._ranks * self.rank_scale_factor


def transform_map_old(fixed, mutable, origin=None, furthest_first=True):
    """Greedy matching (distributions.py:2804-2857): visit the fixed points by distance from
    ``origin`` (furthest first by default) and give each the closest still-unused mutable
    point.  Returns ``mutable`` re-ordered to line up with ``fixed``."""
    fixed, mutable = np.array(_as_np(fixed)), np.array(_as_np(mutable))
    if fixed.shape != mutable.shape:
        raise ValueError("transform_map: both inputs must have exactly the same shape.")
    if origin is None:
        origin = np.zeros(fixed.shape[1])
    elif np.shape(origin)[0] != fixed.shape[1]:
        raise ValueError("transform_map: origin must have the same dimension as fixed.")
    order = np.argsort(np.linalg.norm(fixed - origin, axis=1))
    if furthest_first:
        order = order[::-1]
    out = np.zeros_like(mutable)
    used = np.zeros(mutable.shape[0], dtype=bool)
    for f in order:
        d = np.linalg.norm(fixed[f] - mutable, axis=1)
        d[used] = 2 * np.amax(d)
        pick = np.argmin(d)
        used[pick] = True
        out[f] = mutable[pick]
    return out


def transform_map(fixed, mutable):
    """Minimum-total-distance matching (Hungarian method, scipy) of ``mutable`` onto
    ``fixed`` (distributions.py:2860-2903)."""
    from scipy.optimize import linear_sum_assignment
    fixed, mutable = np.array(_as_np(fixed)), np.array(_as_np(mutable))
    if fixed.shape != mutable.shape:
        raise ValueError("transform_map: both inputs must have exactly the same shape.")
    cost = np.linalg.norm(fixed[:, None, :] - mutable[None, :, :], axis=2)
    rows, cols = linear_sum_assignment(cost)
    return mutable[cols[rows]]


class ImageBasePoints(BasePointDistributionBase):
    """Random points whose count per pixel is the pixel's grey level index
    (distributions.py:2906-3003).  ``filename`` may also be a 2-D array of grey values."""

    def __init__(self, filename, x_size, y_size=None, **kwargs):
        self.x_size = x_size
        self.y_size = y_size or x_size
        raw = _read_grey_image(filename) if isinstance(filename, str) else \
            np.array(_as_np(filename), dtype=np.float64)
        self._x_res, self._y_res = raw.shape
        levels, inverse = np.unique(raw, return_inverse=True)
        self._grey_levels = len(levels)
        self._image = np.reshape(np.arange(self._grey_levels)[inverse.reshape(-1)],
                                 (self._x_res, self._y_res))
        super().__init__(**kwargs)

    x_size = property(lambda self: self._x_size)
    y_size = property(lambda self: self._y_size)
    grey_levels = property(lambda self: self._grey_levels)
    x_res = property(lambda self: self._x_res)
    y_res = property(lambda self: self._y_res)

    @x_size.setter
    def x_size(self, val):
        if float(val) <= 0:
            raise ValueError("ImageBasePoints: x_size must be > 0.")
        self._x_size = float(val)

    @y_size.setter
    def y_size(self, val):
        if float(val) <= 0:
            raise ValueError("ImageBasePoints: y_size must be > 0.")
        self._y_size = float(val)

    def _update(self):
        counts = self._image.reshape(-1)
        total = int(counts.sum())
        x_edges = np.linspace(-self._x_size / 2, self._x_size / 2, self._x_res + 1)
        y_edges = np.linspace(-self._y_size / 2, self._y_size / 2, self._y_res + 1)
        # pixel (row-major) of every point, then a uniform offset inside that pixel
        pix = np.repeat(np.arange(counts.size), counts)
        ix, iy = pix // self._y_res, pix % self._y_res
        u = _uniform(total).cpu().numpy()
        v = _uniform(total).cpu().numpy()
        px = x_edges[ix] + u * (x_edges[ix + 1] - x_edges[ix])
        py = y_edges[iy] + v * (y_edges[iy + 1] - y_edges[iy])
        self._points = _f64(np.stack([px, py], axis=1))

    @BasePointDistributionBase.ranks.setter
    def ranks(self, val):
        self._ranks = val


class PrecompiledBasePoints(RecursivelyUpdatable):
    """Stored base points (and ranks), optionally re-sampled with replacement and jittered at
    each update (distributions.py:3006-3195).  File format: pickle of
    ``{"points": ndarray|None, "ranks": ndarray|None}`` (distributions.py:3080-3095)."""

    def __init__(self, arg, sample_count=100, do_downsample=True, perturbation=None, **kwargs):
        if type(arg) is str:
            import pickle
            with open(arg, "rb") as f:
                data = pickle.load(f)
            self._full_points, self._full_ranks = data["points"], data["ranks"]
        else:
            self._full_points = getattr(arg, "points", None)
            self._full_ranks = getattr(arg, "ranks", None)
        self._full_points = None if self._full_points is None else _f64(_as_np(self._full_points))
        self._full_ranks = None if self._full_ranks is None else _f64(_as_np(self._full_ranks))
        self._points = self._ranks = None
        self.perturbation = perturbation
        self.sample_count = sample_count
        self.do_downsample = do_downsample
        RecursivelyUpdatable.__init__(self, **kwargs)

    def save(self, filename):
        import pickle
        out = {"points": None if self._full_points is None else self._full_points.cpu().numpy(),
               "ranks": None if self._full_ranks is None else self._full_ranks.cpu().numpy()}
        with open(filename, "wb") as f:
            pickle.dump(out, f, pickle.HIGHEST_PROTOCOL)

    def _update(self):
        if self.do_downsample and self.sampling_domain_size > 0:
            n = self.sampling_domain_size
            idx = (_uniform(self.sample_count) * n).long().clamp_(max=n - 1)
            if self._full_points is not None:
                self._points = self._full_points[idx.to(self._full_points.device)]
            if self._full_ranks is not None:
                self._ranks = self._full_ranks[idx.to(self._full_ranks.device)]
        else:
            self._points, self._ranks = self._full_points, self._full_ranks
        if self._perturbation is not None and self._points is not None:
            gen, dev = _generator_for(self._points.device)
            noise = torch.randn(self._points.shape, dtype=torch.float64, generator=gen, device=dev)
            self._points = self._points + noise * self._perturbation

    def clear(self):
        self._full_points = self._full_ranks = self._points = self._ranks = None

    def _generate_update_handles(self):
        return []

    @property
    def sampling_domain_size(self):
        return 0 if self._full_points is None else int(self._full_points.shape[0])

    points = property(lambda self: self._points)
    ranks = property(lambda self: self._ranks)
    full_points = property(lambda self: self._full_points)
    full_ranks = property(lambda self: self._full_ranks)
    perturbation = property(lambda self: self._perturbation)

    @points.setter
    def points(self, val):
        self._full_points = None if val is None else _f64(_as_np(val))

    @ranks.setter
    def ranks(self, val):
        self._full_ranks = None if val is None else _f64(_as_np(val))

    @perturbation.setter
    def perturbation(self, val):
        if val is not None:
            if self._full_points is None:
                raise ValueError("PrecompiledBasePoints: perturbation must be None, scalar, or "
                                 "must have one entry per dimension of the points.")
            try:
                val = _f64(np.broadcast_to(np.asarray(val, dtype=np.float64),
                                           (self._full_points.shape[1],)).copy())
            except ValueError as e:
                raise ValueError("PrecompiledBasePoints: perturbation must be None, scalar, or "
                                 "must have one entry per dimension of the points.") from e
        self._perturbation = val


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
        if isinstance(base, _DeviceRandom):
            # the program of a device-random distribution carries ONE transformation; with a
            # second one the distribution goes back to drawing with torch ops
            base.__dict__.setdefault("_transformations", []).append(self)

    def _apply_transformation(self):
        if self._base.__dict__.get("_device_active"):
            return                    # (part of the distribution's program)
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
