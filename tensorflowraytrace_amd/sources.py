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


def _lib_kinds():
    from . import _lib
    return _lib.SRC_APERTURE, _lib.SRC_POINT, _lib.SRC_ANGULAR


_GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


class DeviceRaySet:
    """The field set of a source whose rays are made by a device program (csrc/tfrt_source.hip):
    a read-only mapping like the reference's source dict (``x_start`` ... ``z_end``,
    ``wavelength``, the extra fields), but nothing is computed until somebody asks -- the trace
    takes ``ray_block(dtype)`` (one launch into a persistent buffer, no float64 field columns, no
    stack / cast), a field is one more launch on first use.  ``permuted(index)`` is the same set
    with ray j = ray ``index[j]`` (a coherent order), made by the same programs: an ordered copy
    of a re-drawn source costs a second launch, not a gather."""

    def __init__(self, source, first=0, count=None, index=None):
        self._src = source
        self._first = int(first)
        self._n = int(source._dev_n if count is None else count)
        self._index = index
        self._index_id = None if index is None else (id(index), index._version)

    # ------------------------------------------------------------------ identity for caches
    @property
    def epoch(self):
        return self._src._dev_epoch

    @property
    def cache_key(self):
        """Changes whenever the rays do (an update of the source re-draws them in place)."""
        return ("device-source", id(self._src), self._src._dev_epoch, self._first, self._n,
                self._index_id)

    @property
    def identity(self):
        """What stays the same from update to update (the buffers are persistent)."""
        return ("device-source", id(self._src), self._first, self._n, self._src._dev_program_key)

    n_rays = property(lambda self: self._n)
    device = property(lambda self: self._src._dev_device)

    def shard(self, lo, hi):
        return self._src._device_view(self._first + int(lo), int(hi) - int(lo), None)

    def permuted(self, index):
        return self._src._device_view(self._first, self._n, index)

    # ------------------------------------------------------------------------- mapping
    def keys(self):
        ks = list(_GEO3)
        if self._src._wavelengths is not None:
            ks.append("wavelength")
        return ks + [f for f in self._src._extra_fields if f not in ks]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, key):
        return key in self.keys()

    def __bool__(self):
        return True

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _memo(self, name, make):
        cache = self._src._dev_cache
        k = (self._first, self._n, self._index_id, name)
        if k not in cache:
            cache[k] = make()
        return cache[k]

    def __getitem__(self, key):
        src = self._src
        if key in _GEO3:
            block = self._memo("__fields__", lambda: src._device_rays(
                self._first, self._n, self._index, None, fields=True)[1])
            return block[_GEO3.index(key)]
        if key == "wavelength" and src._wavelengths is not None:
            return self._memo(key, lambda: self._cut(src._wavelength_column()))
        if key in src._extra_fields:
            return self._memo(key, lambda: src._device_extra(key, self._first, self._n, self._index))
        raise KeyError(key)

    def _cut(self, column):
        """rows [first, first + n) of a natural-order column, through the index if there is one."""
        if column.stride(0) == 0 or column.shape[0] == 1:       # one value for every ray
            return column.expand(self._n, *column.shape[1:]) if column.shape[0] == 1 else \
                column[:self._n]
        part = column[self._first:self._first + self._n]
        if self._index is None:
            return part
        return part.index_select(0, self._index.long())

    def order(self, face_verts=None, out=None, stable=True):
        """A coherent order of the set's rays (ops.ray_order) straight from the program
        (``stable=False``: ops.source3d_order's faster sort, ties in any order)."""
        from . import ops
        src = self._src
        axis = src.axis_hint() if hasattr(src, "axis_hint") else None
        return ops.source3d_order(src._dev_program[1], self._n, self._first, face_verts, axis,
                                  out=out, device=src._dev_device, stable=stable)

    def ray_block(self, dtype):
        """(6, n) block of the ray-state dtype: a persistent buffer, filled once per update."""
        return self._src._device_rays(self._first, self._n, self._index, dtype)[0]


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

    # --------------------------------------------------------------- device programs
    def _device_inputs(self):
        """(kind, a, b, extra) of tfrt_source3d_program for this source, ``a`` / ``b`` the input
        distributions' point tensors or device-random distributions; None: no device form."""
        return None

    def _device_program(self):
        """The source as a tfrt_source3d_program when that is possible: 3-D, undense, on a HIP
        device, at least one input a device-random distribution (a source of static inputs keeps
        its tensors: they are computed once anyway)."""
        from . import _lib
        if (getattr(self, "_dimension", None) != 3 or self.dense or not dist._device_random
                or config.get_device().type != "cuda"):
            return None
        # (a program bakes center, central_angle, the wavelengths and table inputs in as numbers:
        # a source with one of them requiring grad keeps the differentiable torch path)
        baked = [self.__dict__.get("_center"), self.__dict__.get("_central_angle"),
                 self.__dict__.get("_wavelengths")]
        if any(isinstance(t, torch.Tensor) and t.requires_grad for t in baked):
            return None
        spec = self._device_inputs()
        if spec is None:
            return None
        kind, a, b, extra = spec
        if any(isinstance(d, torch.Tensor) and d.requires_grad for d in (a, b)):
            return None
        live = [d for d in (a, b) if isinstance(d, dist._DeviceRandom)
                and d.__dict__.get("_device_active")]
        if not live:
            return None
        counts, parts, keep, key = [], [], [], [kind]
        for d in (a, b):
            if d is None:
                parts.append(None)
                key.append(None)
                continue
            if isinstance(d, dist._DeviceRandom) and d.__dict__.get("_device_active"):
                pg = d.program()
                parts.append(pg)
                counts.append(int(pg.count))
                key.append(("program", id(d), id(pg)))
                continue
            pts = d if isinstance(d, torch.Tensor) else None
            if pts is None or pts.dim() != 2 or pts.shape[1] not in (2, 3):
                return None
            tkey = (id(pts), pts._version)
            memo = self.__dict__.setdefault("_dev_tables", {})
            hit = memo.get(tkey)
            if hit is None:
                t = _f64(pts.detach())
                if t.shape[1] == 2:
                    t = torch.cat([torch.zeros_like(t[:, :1]), t], dim=1)
                if len(memo) > 8:
                    memo.clear()
                hit = memo[tkey] = (pts, t.contiguous())
            pg = _lib.PointsProgram()
            pg.kind, pg.count, pg.table = _lib.PTS_TABLE, hit[1].shape[0], hit[1].data_ptr()
            keep.append(hit[1])
            parts.append(pg)
            counts.append(int(pg.count))
            key.append(("table", tkey))
        n = max(counts)
        if any(c not in (1, n) for c in counts):
            return None
        if self._wavelengths is not None and self._wavelengths.numel() not in (1, n):
            return None
        key.append(extra["key"])
        key = tuple(key)
        cached = self.__dict__.get("_dev_program")
        if cached is not None and cached[0] == key:
            return cached[1], n, key
        sp = _lib.Source3DProgram()
        sp.kind = kind
        sp.swap = 1 if extra.get("swap") else 0
        if parts[0] is not None:
            sp.a = parts[0]
        sp.b = parts[1]
        c = extra.get("center")
        for k in range(3):
            sp.center[k] = 0.0 if c is None else c[k]
        q = extra.get("quat")
        sp.has_quat = 0 if q is None else 1
        for k in range(4):
            sp.quat[k] = 0.0 if q is None else q[k]
        sp.ray_length = float(extra.get("ray_length", 1.0))
        sp.n_rays = n
        self._dev_program = (key, sp, keep, live)
        return sp, n, key

    def _host_values(self, name, tensor, normalise=False):
        """A small tensor's values as python floats, read back once per tensor version."""
        memo = self.__dict__.setdefault("_dev_host", {})
        key = (id(tensor), tensor._version)
        hit = memo.get(name)
        if hit is None or hit[0] != key:
            t = tensor.detach().cpu().double().reshape(-1)
            if normalise:
                t = t / torch.linalg.norm(t)
            hit = memo[name] = (key, t.tolist(), tensor)
        return hit[1]

    def _enter_device(self, sp, n, key):
        from . import ops
        # step the distributions that were updated since they last drew: one launch for all
        pending = []
        for d in self._dev_program[3]:
            pending.extend(d.pending_epochs())
        while pending:
            ops.epoch_advance([c for c, _ in pending])
            pending = [(c, k - 1) for c, k in pending if k > 1]
        self._dev_n, self._dev_program_key = n, key
        self._dev_device = config.get_device()
        self._dev_epoch = self.__dict__.get("_dev_epoch", 0) + 1
        self._dev_cache = {}
        self._domain_sizes = {"whole": n}
        self._domain_gathers = {}
        self._needs_resize = False
        self._fields = self._device_view(0, n, None)
        self._memo_key = None

    def note_external_update(self):
        """The distributions' device counters were stepped without Python (a replayed launch
        graph of an optimiser step): forget what was materialised for earlier draws."""
        if isinstance(self._fields, DeviceRaySet):
            self._dev_epoch += 1
            self._dev_cache = {}
            for d in self._dev_program[3]:
                d._drawn = {}
                d.epoch = d.__dict__.get("epoch", 0) + 1

    def _device_view(self, first, n, index):
        views = self.__dict__.setdefault("_dev_views", {})
        k = (first, n, None if index is None else id(index))
        v = views.get(k)
        if v is None or (index is not None and v._index is not index):
            if len(views) > 16:
                views.clear()
            v = views[k] = DeviceRaySet(self, first, n, index)
        return v

    def _device_rays(self, first, n, index, dtype, fields=False):
        """Generate (once per update and destination) the ray block of ``dtype`` and / or the
        float64 field columns of rays ``first + index[j]``."""
        from . import ops
        sp = self._dev_program[1]
        bufs = self.__dict__.setdefault("_dev_buffers", {})
        ik = None if index is None else (id(index), index._version)
        rays = fl = None
        if dtype is not None:
            bk = (first, n, None if index is None else "index", dtype)
            ent = bufs.get(bk)
            if ent is None or ent[0].device != self._dev_device:
                ent = bufs[bk] = [torch.empty((6, n), dtype=dtype, device=self._dev_device), None, None]
            if ent[1] != (self._dev_epoch, ik):
                ops.source3d_generate(sp, n, first=first, index=index, rays_out=ent[0])
                ent[1], ent[2] = (self._dev_epoch, ik), index
            rays = ent[0]
        if fields:
            _, fl = ops.source3d_generate(sp, n, first=first, index=index, fields=True,
                                          device=self._dev_device)
        return rays, fl

    def _wavelength_column(self):
        w = self._wavelengths
        return w if w.numel() != 1 else w.reshape(1).expand(self._dev_n)

    def _device_extra(self, field, first, n, index):
        """An extra field of rays ``first + index[j]``, resolved when somebody asks."""
        from . import ops
        items = self._extra_fields[field]
        if len(items) == 3 and items[2] == "points" and isinstance(items[1], dist._DeviceRandom) \
                and items[1].__dict__.get("_device_active") and items[1]._sample_total() == self._dev_n:
            # the points of one of the source's own distributions: made in the asked order
            d = items[1]
            if index is None and first == 0 and n == self._dev_n:
                return d._draw("points")
            from . import _lib
            cols = 3 if (d._transformed() or d._kind in (_lib.PTS_SPHERE_UNIFORM,
                                                         _lib.PTS_SPHERE_LAMBERT)) else 2
            d.flush_epoch()
            return ops.points_generate(d.program(), n, first=first, index=index, columns=cols,
                                       device=self._dev_device)[0]
        domain, value = self._resolve_extra(items)
        value = value if isinstance(value, torch.Tensor) else torch.as_tensor(np.asarray(value))
        value = value.to(self._dev_device)
        if value.dim() < 2 and value.shape[:1] != (self._dev_n,):
            value = value.expand(self._dev_n)
        part = value[first:first + n]
        return part if index is None else part.index_select(0, index.long())

    def _update(self):
        prog = self._device_program()
        if prog is not None:
            self._enter_device(*prog)
            return
        if isinstance(self._fields, DeviceRaySet):
            self._fields = {}
            self._memo_key = None
            self._needs_resize = True
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

    def _device_inputs(self):
        ad = self._angular_distribution
        b = ad if isinstance(ad, dist._DeviceRandom) else (ad.angles if hasattr(ad, "angles") else ad.points)
        c = self._host_values("center", self._center)
        q = self._host_values("quat", self._central_angle, normalise=True)
        return _lib_kinds()[1], None, b, dict(
            center=c, quat=q, ray_length=float(self.ray_length), swap=not self.start_on_center,
            key=("point", tuple(c), tuple(q), float(self.ray_length), bool(self.start_on_center)))

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

    def _device_inputs(self):
        ad, bd = self._angular_distribution, self._base_point_distribution
        b = ad if isinstance(ad, dist._DeviceRandom) else (ad.angles if hasattr(ad, "angles") else ad.points)
        a = bd if isinstance(bd, dist._DeviceRandom) else bd.points
        c = self._host_values("center", self._center)
        q = self._host_values("quat", self._central_angle, normalise=True)
        return _lib_kinds()[2], a, b, dict(
            center=c, quat=q, ray_length=float(self.ray_length), swap=not self.start_on_base,
            key=("angular", tuple(c), tuple(q), float(self.ray_length), bool(self.start_on_base)))

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

    def _device_inputs(self):
        sd, ed = self._start_point_distribution, self._end_point_distribution
        a = sd if isinstance(sd, dist._DeviceRandom) else sd.points
        b = ed if isinstance(ed, dist._DeviceRandom) else ed.points
        return _lib_kinds()[0], a, b, dict(key=("aperture",))

    def axis_hint(self):
        """Direction the rays mostly share, when the programs tell (two shifted planar
        distributions): from the middle of the start points to the middle of the end points."""
        prog = self.__dict__.get("_dev_program")
        if prog is None:
            return None
        sp = prog[1]
        if sp.a.kind in (1, 2) and sp.b.kind in (1, 2):
            sa = [sp.a.shift[k] if sp.a.has_shift else 0.0 for k in range(3)]
            sb = [sp.b.shift[k] if sp.b.has_shift else 0.0 for k in range(3)]
            ax = [sb[k] - sa[k] for k in range(3)]
            if sum(v * v for v in ax) > 0.0:
                return ax
        return None

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
