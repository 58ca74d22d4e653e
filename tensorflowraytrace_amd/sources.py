"""
Ray sources (tfrt/sources.py), host side: dict-like sets of per-ray fields
``x_start, y_start[, z_start], x_end, y_end[, z_end], wavelength`` + user extra fields.

A *dense* source emits one ray for every combination of its input domains (meshgrid of the
domains in declaration order, like the reference's ``tf.meshgrid`` of index ranges,
sources.py:242-254); an *undense* source matches its inputs 1:1.  ``extra_fields`` entries are
``name: (domain, value)`` or ``name: (domain, object, attribute)`` (sources.py:157-167,
282-309).

Two fixes relative to the reference at HEAD, which is broken for 2-D rotated sources
(SURVEY.md section 2.3): a 2-D ``center`` is a 2-vector, and 2-D base points are rotated by
the central angle with an ordinary rotation matrix.
"""
import math
import pickle
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import config
from . import distributions as dist
from .update import RecursivelyUpdatable

PI = math.pi


def _f64(x):
    return config.as_f64(x)


class SourceBase(RecursivelyUpdatable, ABC):
    def __init__(self, extra_fields={}, standard_domains=set(), dense=True,
                 always_resize=False, **kwargs):
        self.extra_fields = extra_fields
        self._standard_domains = set(standard_domains) | {"whole", "wavelength"}
        self.dense = dense
        self._fields = {}
        self._domain_gathers = {}
        self._domain_sizes = {}
        self._needs_resize = True
        self.always_resize = always_resize
        super().__init__(**kwargs)

    def _set_dimension(self, dimension):
        if dimension not in {2, 3}:
            raise ValueError("Source: dimension must be 2 or 3")
        self._dimension = dimension

    @staticmethod
    def validate_extra_fields(extra_fields):
        for field, args in extra_fields.items():
            if type(field) is not str:
                raise ValueError("Source extra fields: keys must be strings.")
            if len(args) not in (2, 3):
                raise ValueError(
                    "Source extra fields: every entry must be either a 2-tuple of "
                    "(domain, value), or a 3-tuple of (domain, object, attribute).")

    def make_vars(self, internal_vars):
        out = {}
        for name, (domain, var) in internal_vars.items():
            var = self._as_tensor(name, var)
            if self.dense:
                var = var[self._domain_gathers[domain]]
            elif var.dim() < 2:
                var = var.expand(self._domain_sizes["whole"])
            out[name] = var
        return out

    def _as_tensor(self, name, var):
        """float64 device tensor of an internal variable; python lists / numbers are converted
        once per value (the same constant wavelength list every step then stays one tensor,
        which lets the engine reuse everything derived from it)."""
        if isinstance(var, torch.Tensor):
            return _f64(var)
        try:
            key = (tuple(np.ravel(np.asarray(var, dtype=np.float64)).tolist()), np.shape(var),
                   str(config.get_device()))
        except (TypeError, ValueError):
            return _f64(var)
        memo = self.__dict__.setdefault("_const_tensors", {})
        entry = memo.get(name)
        if entry is None or entry[0] != key:
            entry = memo[name] = (key, _f64(var))
        return entry[1]

    def resize(self):
        self._needs_resize = True

    def _resolve_extra(self, items):
        if len(items) == 2:
            domain, raw = items
        else:
            domain, obj, attrb = items
            try:
                raw = obj[attrb]
            except (TypeError, KeyError, IndexError):
                raw = getattr(obj, attrb)
        value = raw() if callable(raw) else raw
        return domain, value

    def _resize(self):
        self._needs_resize = False
        sizes = {}

        def add(domain, value):
            shape = tuple(torch.as_tensor(value).shape) if not isinstance(value, torch.Tensor) \
                else tuple(value.shape)
            sizes.setdefault(domain, []).append(shape[0] if len(shape) else 1)

        for _name, (domain, value) in self._internal_vars().items():
            add(domain, value)
        for _field, items in self._extra_fields.items():
            domain, value = self._resolve_extra(items)
            add(domain, value)

        self._domain_sizes = {}
        for domain, ss in sizes.items():
            s = set(ss)
            if s == {1}:
                self._domain_sizes[domain] = 1
            else:
                s -= {1}
                if len(s) != 1:
                    raise ValueError("Source resize: found incompatible shapes in the same domain.")
                self._domain_sizes[domain] = s.pop()

        if self.dense:
            domains = [d for d in self._domain_sizes if d != "whole"]
            dev = config.get_device()
            ranges = [torch.arange(self._domain_sizes[d], device=dev) for d in domains]
            if len(ranges) == 1:
                grids = [ranges[0]]
            else:
                grids = torch.meshgrid(*ranges, indexing="xy")
            self._domain_gathers = {d: g.reshape(-1) for d, g in zip(domains, grids)}
            whole = 1
            for d in domains:
                whole *= self._domain_sizes[d]
            self._domain_sizes["whole"] = whole
        else:
            self._domain_gathers = {}
            var_size = 1
            for size in self._domain_sizes.values():
                if size == 1:
                    continue
                if var_size == 1:
                    var_size = size
                if var_size != size:
                    raise ValueError(
                        "Source resize: found incompatibly sized variables with an undense source.")
            self._domain_sizes["whole"] = var_size

    def publish_extra_fields(self):
        for field, items in self._extra_fields.items():
            domain, value = self._resolve_extra(items)
            value = value if isinstance(value, torch.Tensor) else torch.as_tensor(
                np.asarray(value))
            value = value.to(config.get_device())
            if value.dim() < 2:
                value = value.expand(self._domain_sizes[domain])
            if domain != "whole" and self.dense:
                value = value[self._domain_gathers[domain]]
            self[field] = value

    def _update(self):
        if self._needs_resize or self.always_resize:
            self._resize()
        ivars = self._internal_vars()
        # memoize: identical input tensors (same objects, same versions) -> identical fields
        key = tuple((n, d, id(v), getattr(v, "_version", None)) for n, (d, v) in ivars.items())
        key += tuple((f, id(self._resolve_extra(it)[1])) for f, it in self._extra_fields.items())
        key += (self.dense, self._dimension_key())
        if getattr(self, "_memo_key", None) == key:
            return
        self._memo_inputs = [v for _, (_, v) in ivars.items()]  # keep ids unique
        self._internal_update(self.make_vars(ivars))
        self.publish_extra_fields()
        self._memo_key = key

    def _dimension_key(self):
        return (getattr(self, "_dimension", None), getattr(self, "ray_length", None),
                getattr(self, "start_on_center", None), getattr(self, "start_on_base", None),
                id(getattr(self, "_center", None)), id(getattr(self, "_central_angle", None)))

    def snapshot(self, do_update=True):
        if do_update:
            self.update()
        return {f: torch.as_tensor(v).clone() for f, v in self.items()}

    @abstractmethod
    def _internal_vars(self):
        raise NotImplementedError

    @abstractmethod
    def _internal_update(self, expanded_vars):
        raise NotImplementedError

    @property
    def extra_fields(self):
        return self._extra_fields

    @extra_fields.setter
    def extra_fields(self, val):
        self.validate_extra_fields(val)
        self._extra_fields = val

    @property
    def standard_domains(self):
        return self._standard_domains

    @property
    def dimension(self):
        return self._dimension

    def __getitem__(self, key):
        return self._fields[key]

    def __setitem__(self, key, item):
        self._fields[key] = item

    def __bool__(self):
        return bool(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()

    def _set_ray_fields(self, start, end, wavelengths, swap=False):
        if swap:
            start, end = end, start
        names = "xyz"[: self._dimension]
        for i, a in enumerate(names):
            self[a + "_start"] = start[:, i].contiguous()
            self[a + "_end"] = end[:, i].contiguous()
        if wavelengths is not None:
            self["wavelength"] = wavelengths


class ManualSource(SourceBase):
    """A source filled by hand: ``src["x_start"] = ...`` (sources.py:363-382)."""

    def __init__(self, dimension, **kwargs):
        self._set_dimension(dimension)
        super().__init__(**kwargs)

    def __setitem__(self, key, item):
        if not isinstance(item, torch.Tensor):
            arr = np.asarray(item)
            item = torch.as_tensor(arr, dtype=torch.float64 if arr.dtype.kind == "f" else None)
        self._fields[key] = item.to(config.get_device())

    def _generate_update_handles(self):
        return []

    def _internal_vars(self):
        return {}

    def _internal_update(self, expanded_vars):
        pass

    def _resize(self):
        self._needs_resize = False
        n = next((v.shape[0] for v in self._fields.values()), 1)
        self._domain_sizes = {"whole": n}
        self._domain_gathers = {}

    def make_vars(self, internal_vars):
        return {}


class RotationBase:
    """Central-angle handling shared by PointSource and AngularSource (sources.py:386-460):
    2-D: scalar angle; 3-D: ``angle_type`` 'vector' (rotate +x onto the vector) or
    'quaternion'."""

    _x_axis = (1.0, 0.0, 0.0)

    def __init__(self, central_angle, angle_type):
        if angle_type not in ("vector", "quaternion"):
            raise ValueError("Source: angle_type must be 'vector' or 'quaternion'.")
        self._angle_type = angle_type
        self.central_angle = central_angle

    @property
    def central_angle(self):
        return self._central_angle

    @central_angle.setter
    def central_angle(self, val):
        val = _f64(val)
        if self.dimension == 2:
            if val.dim() != 0:
                raise ValueError("Source: central_angle must be scalar.")
            self._central_angle = val
        elif self._angle_type == "vector":
            if tuple(val.shape) != (3,):
                raise ValueError("Source: central_angle must be size (3,).")
            self._central_angle = dist.get_rotation_quaternion_from_u_to_v(self._x_axis, val)
        else:
            if tuple(val.shape) != (4,):
                raise ValueError("Source: central_angle must be size (4,).")
            self._central_angle = val

    def _rotate_angles(self, angles):
        if self._dimension == 2:
            return angles + self._central_angle
        return dist.rotate_vector_by_quaternion(self._central_angle, angles)

    def _rotate_points(self, points):
        if self._dimension == 2:
            c, s = torch.cos(self._central_angle), torch.sin(self._central_angle)
            x, y = points[:, 0], points[:, 1]
            return torch.stack([c * x - s * y, s * x + c * y], dim=1)
        if points.shape[1] == 2:
            points = torch.cat([torch.zeros_like(points[:, :1]), points], dim=1)
        return dist.rotate_vector_by_quaternion(self._central_angle, points)


def _center(val, dimension, who):
    val = _f64(val)
    if tuple(val.shape) != (dimension,):
        raise ValueError(f"{who}: center must be size ({dimension},).")
    return val


class PointSource(SourceBase, RotationBase):
    """Rays leaving one point along an angular distribution (sources.py:464-675)."""

    def __init__(self, dimension, center, central_angle, angular_distribution, wavelengths,
                 start_on_center=True, ray_length=1.0, angle_type="vector", **kwargs):
        self._set_dimension(dimension)
        RotationBase.__init__(self, central_angle, angle_type)
        self.center = center
        self._angular_distribution = angular_distribution
        self._wavelengths = None if wavelengths is None else _f64(wavelengths).reshape(-1)
        self.start_on_center = start_on_center
        self.ray_length = ray_length
        SourceBase.__init__(self, standard_domains={"angle"}, **kwargs)

    def _internal_update(self, ev):
        angles = self._rotate_angles(ev["angles"])
        if self.dimension == 2:
            start = self._center.reshape(1, 2).expand(angles.shape[0], 2)
            end = start + self.ray_length * torch.stack([torch.cos(angles), torch.sin(angles)], 1)
        else:
            start = self._center.reshape(1, 3).expand(angles.shape[0], 3)
            end = start + self.ray_length * angles
        self._set_ray_fields(start, end, ev.get("wavelengths"), swap=not self.start_on_center)

    def _internal_vars(self):
        ad = self._angular_distribution
        angles = ad.angles if hasattr(ad, "angles") else ad.points
        out = {"angles": ("angle", angles)}
        if self._wavelengths is not None:
            out["wavelengths"] = ("wavelength", self._wavelengths)
        return out

    def _generate_update_handles(self):
        return [self._angular_distribution.update]

    center = property(lambda self: self._center)

    @center.setter
    def center(self, val):
        self._center = _center(val, self.dimension, "PointSource")

    angular_distribution = property(lambda self: self._angular_distribution)


class AngularSource(SourceBase, RotationBase):
    """Rays from a set of base points along an angular distribution (sources.py:678-915)."""

    def __init__(self, dimension, center, central_angle, angular_distribution,
                 base_point_distribution, wavelengths, start_on_base=True, ray_length=1.0,
                 angle_type="vector", **kwargs):
        self._set_dimension(dimension)
        RotationBase.__init__(self, central_angle, angle_type)
        self.center = center
        self._angular_distribution = angular_distribution
        self._base_point_distribution = base_point_distribution
        self._wavelengths = None if wavelengths is None else _f64(wavelengths).reshape(-1)
        self.start_on_base = start_on_base
        self.ray_length = ray_length
        SourceBase.__init__(self, standard_domains={"base_point", "angle"}, **kwargs)

    def _internal_update(self, ev):
        angles = self._rotate_angles(ev["angles"])
        base = self._rotate_points(ev["base_points"])
        start = self._center + base
        if self.dimension == 2:
            end = start + self.ray_length * torch.stack([torch.cos(angles), torch.sin(angles)], 1)
        else:
            end = start + self.ray_length * angles
        self._set_ray_fields(start, end, ev.get("wavelengths"), swap=not self.start_on_base)

    def _internal_vars(self):
        ad = self._angular_distribution
        angles = ad.angles if hasattr(ad, "angles") else ad.points
        out = {"angles": ("angle", angles),
               "base_points": ("base_point", self._base_point_distribution.points)}
        if self._wavelengths is not None:
            out["wavelengths"] = ("wavelength", self._wavelengths)
        return out

    def _generate_update_handles(self):
        return [self._angular_distribution.update, self._base_point_distribution.update]

    center = property(lambda self: self._center)

    @center.setter
    def center(self, val):
        self._center = _center(val, self.dimension, "AngularSource")

    angular_distribution = property(lambda self: self._angular_distribution)
    base_point_distribution = property(lambda self: self._base_point_distribution)


class AperatureSource(SourceBase):
    """Rays spanning two absolute point sets (sources.py:918-1095)."""

    def __init__(self, dimension, start_point_distribution, end_point_distribution, wavelengths,
                 **kwargs):
        self._set_dimension(dimension)
        self._start_point_distribution = start_point_distribution
        self._end_point_distribution = end_point_distribution
        self._wavelengths = None if wavelengths is None else _f64(wavelengths).reshape(-1)
        super().__init__(standard_domains={"start_point", "end_point"}, **kwargs)

    def _internal_update(self, ev):
        self._set_ray_fields(ev["start_points"], ev["end_points"], ev.get("wavelengths"))

    def _internal_vars(self):
        out = {"start_points": ("start_point", self._start_point_distribution.points),
               "end_points": ("end_point", self._end_point_distribution.points)}
        if self._wavelengths is not None:
            out["wavelengths"] = ("wavelength", self._wavelengths)
        return out

    def _generate_update_handles(self):
        return [self._start_point_distribution.update, self._end_point_distribution.update]

    start_point_distribution = property(lambda self: self._start_point_distribution)
    end_point_distribution = property(lambda self: self._end_point_distribution)


class PrecompiledSource(RecursivelyUpdatable):
    """A stored ray set, optionally re-sampled / perturbed at each update
    (sources.py:1099-1358).  File format: pickle of
    ``{"dimension", "standard_domains", "fields": {name: ndarray}}`` (sources.py:1174-1181)."""

    def __init__(self, arg, sample_count=100, do_downsample=True, start_perturbation=None,
                 end_perturbation=None, **kwargs):
        if type(arg) is str:
            with open(arg, "rb") as f:
                data = pickle.load(f)
            self._dimension = data["dimension"]
            self._standard_domains = data["standard_domains"]
            self._full_fields = {k: np.asarray(v) for k, v in data["fields"].items()}
        elif type(arg) is not int:
            self._dimension = arg._dimension
            self._standard_domains = arg._standard_domains
            self._full_fields = {k: v.detach().cpu().numpy() for k, v in arg._fields.items()}
        else:
            self._dimension = arg
            self._standard_domains = set()
            self._full_fields = {}
        self.start_perturbation = start_perturbation
        self.end_perturbation = end_perturbation
        self._fields = {}
        self.sample_count = sample_count
        self.do_downsample = do_downsample
        RecursivelyUpdatable.__init__(self, **kwargs)

    @property
    def sampling_domain_size(self):
        f = self._full_fields.get("x_start")
        return 0 if f is None else f.shape[0]

    def save(self, filename):
        out = {"dimension": self._dimension, "standard_domains": self._standard_domains,
               "fields": {k: v.detach().cpu().numpy() for k, v in self._fields.items()}}
        with open(filename, "wb") as f:
            pickle.dump(out, f, pickle.HIGHEST_PROTOCOL)

    def _update(self):
        dev = config.get_device()
        n = self.sampling_domain_size
        if self.do_downsample and n > 0:
            idx = (dist._uniform(self.sample_count) * n).long().clamp_(max=n - 1).cpu().numpy()
        else:
            idx = None
        for field, item in self._full_fields.items():
            arr = item if idx is None else item[idx]
            self._fields[field] = torch.as_tensor(arr, device=dev)
        for pert, suffix in ((self.start_perturbation, "_start"), (self.end_perturbation, "_end")):
            if pert is None:
                continue
            p = np.broadcast_to(np.asarray(pert, dtype=np.float64), (self._dimension,))
            for a, sd in zip("xyz"[: self._dimension], p):
                f = self._fields[a + suffix]
                noise = torch.randn(f.shape, dtype=torch.float64, generator=dist._generator
                                    if dist._generator is not None else None).to(dev)
                self._fields[a + suffix] = f + float(sd) * noise

    def _generate_update_handles(self):
        return []

    dimension = property(lambda self: self._dimension)

    def __getitem__(self, key):
        return self._fields[key]

    def __setitem__(self, key, item):
        self._fields[key] = item

    def __bool__(self):
        return bool(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()
