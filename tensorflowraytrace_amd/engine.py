"""
Optical systems and the trace engine (tfrt/engine.py), host side.

``OpticalSystem2D/3D`` collect sources and boundaries and merge them in the reference's order
(optical, stop, target; engine.py:971-1018).  ``OpticalEngine.ray_trace`` hands the merged
scene to ONE fused HIP trace (``ops.trace3d`` / ``ops.trace2d`` -> tfrt_trace*_forward): all
passes run on the device without host round trips, and the ray sets the reference exposes
(``finished_rays``, ``active_rays``, ``stopped_rays``, ``dead_rays``, ``all_rays``) are
materialised lazily from the compacted outputs.  Geometric fields come straight from the
kernels; every other field of an output ray (``wavelength`` and the user's inherited extra
fields, engine.py:2242-2281) is a gather of the source's field by the source-ray index the
kernels carry along -- exact for ``StandardReaction``, which emits one child per active ray.

The result tensors are differentiable w.r.t. parametric boundary parameters and source ray
coordinates: ``torch.autograd`` plays the role of ``tf.GradientTape`` and calls the
hand-derived HIP reverse sweep.
"""
import math
from abc import ABC, abstractmethod

import torch

from . import config, ops, _lib
from . import distributed as tdist
from . import operation as op
from .boundaries import amalgamate as _amalgamate_plain
from .update import RecursivelyUpdatable

OPTICAL = 0
STOP = 1
TARGET = 2

SEGMENT_GEO_SIG = {"x_start", "y_start", "x_end", "y_end"}
ARC_GEO_SIG = {"x_center", "y_center", "angle_start", "angle_end", "radius"}
SOURCE_3D_SIG = {"x_start", "y_start", "z_start", "x_end", "y_end", "z_end"}
TRIANGLE_GEO_SIG = {"xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2", "norm"}

PI = math.pi
_GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
_GEO2 = ("x_start", "y_start", "x_end", "y_end")
_CLASSES = ("active", "finished", "stopped", "dead")
_RESTORE_CHUNK = 4096      # csrc/tfrt_order.hip SCAN_CHUNK: (pass, 32-ray word) pairs per scan chunk


class ReadOnlySet:
    """Read-only view of a field dict (engine.py:27-46)."""

    def __init__(self, fields, empty=None):
        self._fields = fields
        # A class that a trace left without rays is an empty dict in the reference (no key at all,
        # engine.py:1379-1403 amalgamates nothing).  `empty` (field -> zero-length tensor) keeps
        # that -- the set is falsy, lists no keys -- but lets an error function written as
        # ``stack([fin["y_end"], ...])`` run on it: with the rays sharded over ranks a rank whose
        # shard finishes nothing must still take part in the step (its term is an empty sum).
        self._empty = empty or {}

    def __getitem__(self, key):
        try:
            return self._fields[key]
        except KeyError as e:
            if not self._fields and key in self._empty:
                return self._empty[key]
            raise KeyError(f"key {key} not in the signature of this set.") from e

    def __bool__(self):
        return bool(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()


class LazyFields:
    """dict-like ray set whose non-geometric fields (inherited from the source rays by
    source-ray index) are gathered on first access."""

    def __init__(self, ready, lazy):
        self._ready = dict(ready)
        self._lazy = dict(lazy)  # name -> zero-argument callable

    def keys(self):
        return list(self._ready.keys()) + [k for k in self._lazy if k not in self._ready]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, key):
        return key in self._ready or key in self._lazy

    def __bool__(self):
        return bool(self._ready) or bool(self._lazy)

    def __getitem__(self, key):
        if key not in self._ready:
            self._ready[key] = self._lazy[key]()
        return self._ready[key]

    def __setitem__(self, key, value):
        self._ready[key] = value

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default


def amalgamate(stuff, signature=None):
    """Join a list of field sets into one dict by concatenation (engine.py:50-76)."""
    items = [s for s in stuff if bool(s)]
    if len(items) == 1 and not signature:
        if isinstance(items[0], LazyFields):
            return items[0]
        lazy = getattr(items[0], "_fields", None)
        if hasattr(lazy, "ray_block"):      # a source made by a device program (sources.DeviceRaySet)
            return lazy
        return {f: items[0][f] for f in items[0].keys()}
    return _amalgamate_plain(stuff, signature)


def recursive_dict_key_print(dict_in, spacer=""):
    """engine.py:80-99."""
    if type(dict_in) is not dict:
        return
    for key, value in dict_in.items():
        try:
            print(spacer, f"{key} : {tuple(value.shape)}")
        except AttributeError:
            print(spacer, key)
        recursive_dict_key_print(value, spacer + "    ")


def annotation_helper(parent, field, value, valid_shape_field, dtype=torch.float64):
    """Keep ``parent[field]`` populated after every update of ``parent``
    (engine.py:103-142): ``value`` is broadcast to the shape of
    ``parent[valid_shape_field]`` (or called with ``(shape, dtype)`` if callable)."""
    if dtype in (int, "int64"):
        dtype = torch.int64

    if callable(value):
        def f():
            shape = parent[valid_shape_field].shape
            parent[field] = value(shape, dtype)
    else:
        def f():
            ref = parent[valid_shape_field]
            parent[field] = torch.as_tensor(value, dtype=dtype, device=ref.device).expand(
                ref.shape).clone()
    parent.post_update_handles.append(f)


# =================================================================================== systems

class OpticalSystemBase(RecursivelyUpdatable, ABC):
    """Holds sources and boundaries (engine.py:146-250).

    ``intersect_epsilion`` / ``size_epsilion`` / ``ray_start_epsilion`` keep the reference's
    meaning and defaults (1e-10) and are applied in the kernels' float64 decision stage.
    """

    _boundary_sets = ()

    def __init__(self, manual_update_management=False, intersect_epsilion=1e-10,
                 size_epsilion=1e-10, ray_start_epsilion=1e-10, **kwargs):
        self._sources = []
        self._read_only = {}
        self.source_handles = []
        self._amalgamated_sources = {}
        self.manual_update_management = manual_update_management
        self.materials = []
        self.intersect_epsilion = intersect_epsilion
        self.size_epsilion = size_epsilion
        self.ray_start_epsilion = ray_start_epsilion
        self.projection_results = {}
        for name in self._boundary_sets:
            setattr(self, "_" + name, [])
            setattr(self, name + "_handles", [])
            setattr(self, "_amalgamated_" + name, {})
        self._scene_cache = None
        super().__init__(**kwargs)

    @property
    @abstractmethod
    def dimension(self):
        raise NotImplementedError

    def refresh_update_handles(self):
        if not self.manual_update_management:
            self.update_handles = self._generate_update_handles()

    def _generate_update_handles(self):
        handles = list(getattr(self, "source_handles", []))
        for name in self._boundary_sets:
            handles += getattr(self, name + "_handles", [])
        return handles

    def clear_read_only(self):
        self._read_only = {}

    def _ro(self, name, fields):
        if name not in self._read_only:
            self._read_only[name] = ReadOnlySet(fields)
        return self._read_only[name]

    @property
    def sources(self):
        return self._ro("sources", self._amalgamated_sources)

    @sources.setter
    def sources(self, new):
        self.source_handles = []
        for each in new:
            assert each.dimension == self.dimension
            self.source_handles.append(each.update)
        self._sources = new
        self.refresh_update_handles()

    @property
    def materials(self):
        return self._materials

    @materials.setter
    def materials(self, val):
        assert type(val) is list
        self._materials = val

    def _get_set(self, name):
        return self._ro(name, getattr(self, "_amalgamated_" + name))

    def _set_set(self, name, new):
        handles = []
        for each in new:
            assert each.dimension == self.dimension
            handles.append(each.update)
        setattr(self, name + "_handles", handles)
        setattr(self, "_" + name, new)
        self._scene_cache = None
        self.refresh_update_handles()

    def _update(self):
        if bool(self._sources):
            self._amalgamated_sources = amalgamate(self._sources)
        for name in self._boundary_sets:
            lst = getattr(self, "_" + name)
            if bool(lst):
                setattr(self, "_amalgamated_" + name, _LazyAmalgam(lst))
        self._merge_boundaries()
        self.clear_read_only()

    @abstractmethod
    def _merge_boundaries(self):
        raise NotImplementedError

    # n(lambda) for every material x source ray, float64 (operation.py:261-272)
    def material_table(self, wavelength):
        mats = []
        for m in self._materials:
            f = m["n"] if isinstance(m, dict) else m
            mats.append(f(wavelength.to(torch.float64)))
        return torch.stack(mats).contiguous()


class _LazyAmalgam:
    """dict-like concatenation of a list of boundaries, field by field, on demand."""

    def __init__(self, items):
        self._items = [i for i in items if bool(i)]
        self._cache = {}

    def keys(self):
        ks = None
        for i in self._items:
            k = set(i.keys())
            ks = k if ks is None else (ks & k)
        return ks or set()

    def __bool__(self):
        return bool(self._items)

    def __contains__(self, key):
        return key in self.keys()

    def __getitem__(self, key):
        if key not in self._cache:
            parts = [i[key] for i in self._items]
            self._cache[key] = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        return self._cache[key]

    def __setitem__(self, key, value):
        self._cache[key] = value

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def count(self, shape_field):
        return sum(int(i[shape_field].shape[0]) for i in self._items)


def _prop(name):
    return property(lambda self: self._get_set(name), lambda self, new: self._set_set(name, new))


class OpticalSystem3D(OpticalSystemBase):
    """engine.py:871-1166."""

    _boundary_sets = ("optical", "stop", "target")
    optical = _prop("optical")
    stops = _prop("stop")
    targets = _prop("target")

    @property
    def dimension(self):
        return 3

    def update(self):
        """update.py:52-64, with one difference in HOW: the parametric boundaries hand their
        face updates to a batch while the handles run, and the merge below runs them -- and the
        copy of the fixed boundaries into the merged block -- as ONE launch (a system's update is
        a chain of small dependent launches; each costs ~4.5 us whatever it does)."""
        if self.frozen:
            return
        batch = ops.ParamFacesBatch()
        with batch:
            if self.recursively_update and bool(self.update_handles):
                for handle in self.update_handles:
                    handle()
        self.__dict__["_faces_batch"] = batch
        try:
            self._update()
        finally:
            self.__dict__["_faces_batch"] = None
            batch.flush()     # (nothing merged: e.g. no boundary sets yet)
        for handle in self.post_update_handles:
            handle()

    def _merge_boundaries(self):
        """Label and merge optical, stop, target (engine.py:971-1018).  The merged geometry
        is one (M,9) float64 tensor; catagory / material columns are cached int32."""
        sets = [(getattr(self, "_" + n), c) for n, c in
                (("optical", OPTICAL), ("stop", STOP), ("target", TARGET))]
        batch = self.__dict__.get("_faces_batch")
        if batch is not None and batch.requests:
            pending = lambda b: getattr(b, "__dict__", {}).get("_faces_pending") is batch
            batch.flush([b for lst, _ in sets for b in lst if pending(b) or bool(b)])
        parts, cats, grads = [], [], []
        for lst, cat in sets:
            for b in lst:
                if not bool(b):
                    continue
                fv = b.face_verts
                parts.append(fv)
                cats.append((cat, fv.shape[0], b))
                grads.append(fv.requires_grad)
        self._optical_count = sum(n for c, n, _ in cats if c == OPTICAL)
        self._stop_count = sum(n for c, n, _ in cats if c == STOP)
        self._target_count = sum(n for c, n, _ in cats if c == TARGET)
        if not parts:
            self._merged = {}
            self._merged_face_verts = None
            return
        block = batch.merged if batch is not None else None
        if (block is not None and len(block[1]) == len(cats)
                and all(rb is b and r1 - r0 == n for (rb, r0, r1), (_, n, b) in zip(block[1], cats))):
            self._merged_face_verts = block[0]   # (every boundary's rows are already in place)
        else:
            self._merged_face_verts = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        self._merged = _MergedTriangles(self)
        key = tuple((c, n, id(b)) + tuple(
            (id(b[f]), b[f]._version) for f in ("mat_in", "mat_out", "n_in", "n_out") if f in b)
            for c, n, b in cats) + (tuple(grads),)
        if self._scene_cache is None or self._scene_cache[0] != key:
            dev = self._merged_face_verts.device
            catagory = torch.cat([torch.full((n,), c, dtype=torch.int32) for c, n, _ in cats]).to(dev)

            def col(field, dtype):
                if not any(c == OPTICAL and field in b for c, n, b in cats):
                    return None
                out = []
                for c, n, b in cats:
                    if c == OPTICAL and field in b:
                        out.append(b[field].to(device=dev, dtype=dtype).reshape(-1))
                    elif c == OPTICAL:
                        raise KeyError(f"optical boundary lacks field {field}")
                    else:
                        out.append(torch.zeros(n, dtype=dtype, device=dev))
                return torch.cat(out).contiguous()

            gmask = torch.cat([torch.full((n,), 1 if g else 0, dtype=torch.uint8)
                               for (c, n, _), g in zip(cats, grads)]).to(dev)
            # the key holds ids and versions: keep the keyed objects alive next to it, so that a
            # freed tensor's id cannot come back as a different tensor under the same key
            held = [b for _, _, b in cats] + [
                b[f] for _, _, b in cats for f in ("mat_in", "mat_out", "n_in", "n_out") if f in b]
            self._scene_cache = (key, dict(
                catagory=catagory, mat_in=col("mat_in", torch.int32),
                mat_out=col("mat_out", torch.int32), n_in=col("n_in", torch.float64),
                n_out=col("n_out", torch.float64), face_grad_mask=gmask), held)

    def scene_signature(self):
        """What a captured launch sequence has baked in about the scene besides the face tensor:
        which boundaries (identity), their material columns (identity + version), the epsilons
        and the material list.  Cheap (no tensor work): checked before every graph replay."""
        sig = []
        for name in ("_optical", "_stop", "_target"):
            for b in getattr(self, name):
                sig.append((id(b),) + tuple(
                    (id(b[f]), b[f]._version) for f in ("mat_in", "mat_out", "n_in", "n_out")
                    if f in b))
        return (tuple(sig), self.intersect_epsilion, self.size_epsilion,
                self.ray_start_epsilion, tuple(id(m) for m in self.materials))

    def scene_args(self, n_table, index_mode, ghost=False, cluster=False, deterministic=False):
        s = self._scene_cache[1]
        order = None
        if cluster:
            # spatial face order for the sphere hierarchy; computed once per scene topology
            # (faces move a little every step, the clusters' bounding spheres are recomputed
            # from the current vertices inside every trace, so a stale order is still exact).
            # Keyed on the boundary objects and face counts only, not on material fields.
            topo = tuple(self._scene_cache[0][i][:3] for i in range(len(self._scene_cache[0]) - 1))
            cached = getattr(self, "_cluster_cache", None)
            if cached is None or cached[0] != topo:
                cached = (topo, ops.cluster_order(self._merged_face_verts))
                self._cluster_cache = cached
            order = cached[1]
        kw = dict(intersect_epsilion=self.intersect_epsilion, size_epsilion=self.size_epsilion,
                  ray_start_epsilion=self.ray_start_epsilion, face_grad_mask=s["face_grad_mask"],
                  cluster_order=order, deterministic=bool(deterministic))
        # the argument object (and the ctypes struct it caches) only depends on tensors that stay
        # the same from step to step; the face tensor is passed separately to every trace
        memo_key = (id(s), id(n_table), bool(index_mode), bool(ghost), id(order),
                    kw["deterministic"],
                    self.intersect_epsilion, self.size_epsilion, self.ray_start_epsilion)
        memo = getattr(self, "_scene_args_memo", None)
        if memo is not None and memo[0] == memo_key:
            memo[1].face_verts = self._merged_face_verts
            return memo[1]
        args = self._build_scene_args(s, n_table, index_mode, ghost, kw)
        self._scene_args_memo = (memo_key, args, s, n_table, order)  # keep the ids alive
        return args

    def _build_scene_args(self, s, n_table, index_mode, ghost, kw):
        if ghost:
            ones = torch.ones(s["catagory"].shape[0], dtype=torch.float64, device=s["catagory"].device)
            return ops.Scene3DArgs(self._merged_face_verts, s["catagory"], n_in=ones, n_out=ones, **kw)
        if index_mode:
            if s["mat_in"] is None or s["mat_out"] is None:
                raise RuntimeError("StandardReaction('index') needs mat_in / mat_out on every "
                                   "optical boundary")
            return ops.Scene3DArgs(self._merged_face_verts, s["catagory"], mat_in=s["mat_in"],
                                   mat_out=s["mat_out"], n_table=n_table, **kw)
        if s["n_in"] is None or s["n_out"] is None:
            raise RuntimeError("StandardReaction('value') needs n_in / n_out on every optical "
                               "boundary")
        return ops.Scene3DArgs(self._merged_face_verts, s["catagory"], n_in=s["n_in"],
                               n_out=s["n_out"], **kw)

    def intersect(self, rays):
        """Nearest triangle per ray (engine.py:1020-1078), via tfrt_intersect3d."""
        result = {}
        if self._merged_face_verts is not None:
            # the seam keeps the caller's precision (the reference's is float64 throughout)
            block = torch.stack([rays[f] for f in _GEO3])
            if block.dtype not in (torch.float32, torch.float64, torch.float16):
                block = block.to(torch.float64)
            (result["x"], result["y"], result["z"], result["valid"], result["ray_u"],
             result["trig_u"], result["trig_v"], result["gather_trig"]) = ops.intersect3d(
                block, self._merged_face_verts.detach(), self.intersect_epsilion,
                self.size_epsilion, self.ray_start_epsilion)
            result["gather_ray"] = torch.arange(block.shape[1], device=block.device)
            result["gather_trig"] = result["gather_trig"].long()
            result["norm"] = self._merged["norm"][result["gather_trig"]]
        return result

    @staticmethod
    def _intersection(rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2,
                      intersect_epsilion, size_epsilion, ray_start_epsilion):
        """Same signature and returns as engine.py:1103-1166."""
        rays = torch.stack([rx1, ry1, rz1, rx2, ry2, rz2])
        fv = torch.stack([xp, yp, zp, x1, y1, z1, x2, y2, z2], dim=1)
        x, y, z, valid, ray_u, trig_u, trig_v, gather = ops.intersect3d(
            rays, fv, intersect_epsilion, size_epsilion, ray_start_epsilion)
        gather_ray = torch.arange(rays.shape[1], device=rays.device)
        return x, y, z, valid, ray_u, trig_u, trig_v, gather_ray, gather.long()


class _MergedTriangles:
    """The reference's ``system._merged`` dict (TRIANGLE_GEO_SIG + catagory), as views."""

    _cols = {"xp": 0, "yp": 1, "zp": 2, "x1": 3, "y1": 4, "z1": 5, "x2": 6, "y2": 7, "z2": 8}

    def __init__(self, system):
        self._s = system

    def __bool__(self):
        return self._s._merged_face_verts is not None

    def keys(self):
        return set(self._cols) | {"norm", "catagory"}

    def __getitem__(self, key):
        s = self._s
        if key in self._cols:
            return s._merged_face_verts[:, self._cols[key]]
        if key == "catagory":
            return s._scene_cache[1]["catagory"].long()
        if key == "norm":
            lists = s._optical + s._stop + s._target
            return torch.cat([b["norm"] for b in lists if bool(b)], 0)
        raise KeyError(key)


class OpticalSystem2D(OpticalSystemBase):
    """engine.py:254-866: separate segment and arc sets per catagory."""

    _boundary_sets = ("optical_segments", "optical_arcs", "stop_segments", "stop_arcs",
                      "target_segments", "target_arcs")
    optical_segments = _prop("optical_segments")
    optical_arcs = _prop("optical_arcs")
    stop_segments = _prop("stop_segments")
    stop_arcs = _prop("stop_arcs")
    target_segments = _prop("target_segments")
    target_arcs = _prop("target_arcs")

    @property
    def dimension(self):
        return 2

    def _merge_kind(self, kind, geo):
        sets = [(getattr(self, f"_amalgamated_{c}_{kind}"), cat) for c, cat in
                (("optical", OPTICAL), ("stop", STOP), ("target", TARGET))]
        geos, cats, extra = [], [], {f: [] for f in ("mat_in", "mat_out", "n_in", "n_out")}
        for s, cat in sets:
            if not bool(s):
                continue
            g = torch.stack([s[f].to(torch.float64) for f in geo], dim=1)
            n = g.shape[0]
            geos.append(g)
            cats.append(torch.full((n,), cat, dtype=torch.int32, device=g.device))
            for f in extra:
                if cat == OPTICAL and f in s:
                    extra[f].append(s[f].reshape(-1))
                else:
                    extra[f].append(None if cat == OPTICAL else torch.zeros(n, device=g.device))
        if not geos:
            return None
        out = {"geo": torch.cat(geos, 0).contiguous(), "cat": torch.cat(cats).contiguous()}
        for f, parts in extra.items():
            if any(p is None for p in parts) or not parts:
                out[f] = None
            else:
                dt = torch.int32 if f.startswith("mat") else torch.float64
                out[f] = torch.cat([p.to(dt) for p in parts]).contiguous()
        return out

    def _merge_boundaries(self):
        self._merged_segments = self._merge_kind("segments", _GEO2)
        self._merged_arcs = self._merge_kind(
            "arcs", ("x_center", "y_center", "angle_start", "angle_end", "radius"))

    def scene_args(self, n_table, index_mode, ghost=False, finite_tir_gradient=False):
        return ops.Scene2DArgs(self._merged_segments, self._merged_arcs, n_table, index_mode,
                               ghost, self.intersect_epsilion, self.size_epsilion,
                               self.ray_start_epsilion, finite_tir_gradient=finite_tir_gradient)

    @staticmethod
    def _segment_intersection(rx1, ry1, rx2, ry2, sx1, sy1, sx2, sy2, intersect_epsilion,
                              size_epsilion, ray_start_epsilion):
        """engine.py:688-749."""
        rays = torch.stack([rx1, ry1, rx2, ry2])
        seg = torch.stack([sx1, sy1, sx2, sy2], dim=1)
        x, y, valid, ray_u, seg_u, gather = ops.segment_intersection(
            rays, seg, intersect_epsilion, size_epsilion, ray_start_epsilion)
        return x, y, valid, ray_u, seg_u, torch.arange(rays.shape[1], device=rays.device), gather.long()

    @staticmethod
    def _arc_intersection(rx1, ry1, rx2, ry2, xc, yc, a1, a2, r, intersect_epsilion,
                          size_epsilion, ray_start_epsilion):
        """engine.py:768-866."""
        rays = torch.stack([rx1, ry1, rx2, ry2])
        arc = torch.stack([xc, yc, a1, a2, r], dim=1)
        x, y, valid, ray_u, arc_u, gather = ops.arc_intersection(
            rays, arc, intersect_epsilion, size_epsilion, ray_start_epsilion)
        return x, y, valid, ray_u, arc_u, torch.arange(rays.shape[1], device=rays.device), gather.long()


# ==================================================================================== engine

class OpticalEngine:
    """Builds and runs the trace (engine.py:1170-2330).

    Constructor arguments keep the reference's names and defaults.  ``ray_dtype`` (extra)
    selects the ray-state precision inside the kernels (default: ``config.get_ray_dtype()``,
    float32); decisions and Snell math are float64 either way.
    """

    def __init__(self, dimension, operations, optical_system=None,
                 compile_technical_intersections=False, compile_stopped_rays=False,
                 compile_dead_rays=False, compile_finished_rays=True, compile_active_rays=True,
                 dead_ray_length=None, compile_geometry_specific_result=False,
                 new_ray_length=1.0, simple_ray_inheritance={"wavelength"}, ray_dtype=None,
                 ray_shard="auto", accelerate="auto", deterministic=False,
                 finite_tir_gradient=False, coherent="auto", in_place="auto"):
        if dimension not in (2, 3):
            raise ValueError(f"RayEngine: dimension must be 2 or 3, but was given {dimension}.")
        self._dimension = dimension
        self._check_exclusions(operations)
        self._operations = operations
        self._optical_system = optical_system
        self.compile_technical_intersections = compile_technical_intersections
        self.compile_stopped_rays = compile_stopped_rays
        self.compile_dead_rays = compile_dead_rays
        self.compile_finished_rays = compile_finished_rays
        self.compile_active_rays = compile_active_rays
        self.dead_ray_length = dead_ray_length
        self.compile_geometry_specific_result = compile_geometry_specific_result
        self.new_ray_length = new_ray_length
        self.ray_dtype = ray_dtype
        # (rank, world_size): trace only this rank's contiguous block of the source rays;
        # "auto" = follow torch.distributed when a process group is up; None = all rays.
        self.ray_shard = ray_shard
        # How a 3-D trace culls ray-face pairs before the exact float64 decision.  Every mode
        # gives identical results (all filters are conservative):
        #   False / "all-pairs": every pair goes through the float32 bounding-sphere filter;
        #   True / "group": sphere hierarchy over k-d face clusters (superclusters of 8 clusters of
        #            16 faces), rays in their natural order;
        #   "auto" (default): "group" once the merged scene has >= 64 faces.
        # ("sort" -- clusters + Morton-sorted rays -- was slower than "group" on every workload
        # measured and is gone; the name is still accepted and means "group".)
        if accelerate not in (False, True, None, "all-pairs", "group", "sort", "auto"):
            raise ValueError(f"OpticalEngine: unknown accelerate mode {accelerate!r}")
        self.accelerate = accelerate
        # True: the reverse sweep sums face gradients in an order-independent way (scaled 64-bit
        # integers): gradients are bit-identical from run to run (tfrt_scene3d.deterministic);
        # default False: float64 atomics, whose last bits depend on the arrival order.
        self.deterministic = bool(deterministic)
        # 3-D hierarchy mode: trace the rays in a coherent order (ops.ray_order: a Hilbert-curve
        # order of their lines, computed once per source on the device) so that wavefronts of 64
        # neighbouring rays share one walk of the face hierarchy, and bring every ray set back to
        # the reference's order afterwards (tfrt_restore_order) -- invisible to the caller.  The
        # order is made on the device (tfrt_ray_order: ~0.1 ms per million rays, no host sync), so
        # a source re-drawn every step is ordered every step; a static source once.  "auto"
        # (default): every trace of >= 4096 rays, ray_trace() included, until a source shows that
        # its wavefronts are no narrow bundles (more than 5 % of the wavefront-passes left to the
        # grouped kernel: a light guide after a few bounces) -- that source then goes back to
        # natural order.  True: always; False: never.
        if coherent not in ("auto", True, False):
            raise ValueError(f"OpticalEngine: coherent must be 'auto', True or False, got {coherent!r}")
        self.coherent = coherent
        # Coherent traces only.  "auto" / True (default): once a source has shown that its
        # wavefronts are narrow bundles (a trace of it left none to the grouped kernel), all passes
        # of its later traces run in ONE launch with every ray kept in its slot
        # (tfrt_scene3d.in_place: no per-pass compaction, scan or reaction launch); the ray sets
        # are compacted into the reference's order afterwards, when they are asked for.  Results
        # do not depend on it.  False: always the per-pass launch sequence.
        if in_place not in ("auto", True, False):
            raise ValueError(f"OpticalEngine: in_place must be 'auto', True or False, got {in_place!r}")
        self.in_place = in_place
        # 2-D only.  False (default, the reference): a totally reflected ray has a NaN gradient
        # (tf.asin in the unselected tf.where branch, geometry.py:640-646) which poisons every
        # boundary entry it touched; SGD_Optimizer zeroes those (optimizer.py:226-229).  True:
        # the reflect branch's finite gradient instead (tfrt_scene2d.finite_tir_gradient).
        self.finite_tir_gradient = bool(finite_tir_gradient)
        # When True, ray_trace() does not wait for the per-class ray counts: it cuts the output
        # sets with the counts of the previous trace of the same shape and leaves the check to
        # verify_trace() (SGD_Optimizer does this; a wrong guess only costs a re-evaluation of
        # the error function).  Off by default: ray_trace() then returns exact sets.
        self.speculative_counts = False
        # ray_trace() of a source that is traced in place returns once the trace is enqueued and
        # cuts the ray sets when they are first asked for (exact sets, no speculation; False:
        # always wait for the counts inside ray_trace())
        self.lazy_ray_sets = True
        self._predicted = None
        self._pending_trace = None
        self._last_trace = None
        self.clear_ray_history()
        self.last_projection_result = {}

        self.input_signature = set()
        self.output_signature = set()
        self.optical_signature = set()
        self.stop_signature = set()
        self.target_signature = set()
        self.material_signature = set()
        self.simple_ray_inheritance = set(simple_ray_inheritance)
        for o in operations:
            self.input_signature |= o.input_signature
            self.output_signature |= o.output_signature
            self.optical_signature |= o.optical_signature
            self.stop_signature |= o.stop_signature
            self.target_signature |= o.target_signature
            self.material_signature |= o.material_signature
            self.simple_ray_inheritance |= o.simple_ray_inheritance

    # ------------------------------------------------------------------ configuration
    def add_inheritable_field(self, fields):
        if type(fields) is str:
            fields = {fields}
        self.simple_ray_inheritance = self.simple_ray_inheritance | set(fields)

    def _check_exclusions(self, operations):
        exclusions, used = set(), set()
        for o in operations:
            used.add(o.__class__)
            exclusions |= o.exclusions
        if used & exclusions:
            raise RuntimeError(f"RayEngine: discovered exclusive operations: {used & exclusions}")
        self.operations = operations

    def _reaction(self):
        """(index_mode, ghost) of the fused reaction declared by the operations.  With custom
        operations only (their ``main`` makes the new rays, see ``_custom_ops``) the projection
        still runs in the kernels, as a ghost pass whose own children are discarded."""
        for o in self._operations:
            if getattr(o, "ghost", False) and o.active:
                return False, True
            if isinstance(o, op.StandardReaction) and o.active:
                return o.refractive_index_type == "index", False
        if self._custom_ops():
            return False, True
        raise RuntimeError(
            "OpticalEngine: no active reaction operation (StandardReaction / GhostThrough, or an "
            "operation with its own main())")

    def _custom_ops(self):
        """Active operations that bring their own ``main`` (operation.py:25-160): run in Python
        after every projection, like the reference does for all operations."""
        return [o for o in self._operations
                if o.active and not getattr(o, "fused", False)
                and type(o).main is not op.RayOperation.main]

    def _fused_op(self):
        return next((o for o in self._operations if getattr(o, "fused", False) and o.active), None)

    def update(self):
        if self._optical_system is not None:
            self._optical_system.update()

    def annotate(self, op_list=None):
        if bool(self.optical_system):
            for o in (self._operations if op_list is None else op_list):
                o.annotate(self)
        else:
            print("No optical system found, so annotating nothing.")

    dimension = property(lambda self: self._dimension)

    @property
    def new_ray_length(self):
        return self._new_ray_length

    @new_ray_length.setter
    def new_ray_length(self, val):
        self._new_ray_length = float(val)

    @property
    def optical_system(self):
        return self._optical_system

    @optical_system.setter
    def optical_system(self, val):
        if val.dimension != self.dimension:
            raise ValueError(
                f"OpticalEngine: attempted to set an optical system with dimension "
                f"{val.dimension}, but this engine is set to dimension {self.dimension}")
        self._optical_system = val

    # ------------------------------------------------------------------------ validation
    def validate_system(self):
        """Key-set checks of engine.py:1416-1522."""
        system = self.optical_system
        if not bool(system):
            print("No optical system found, so validating nothing.")
            return
        for material in system.materials:
            sig = set(material.keys()) if isinstance(material, dict) else {"n"}
            if not (sig >= self.material_signature):
                raise RuntimeError(
                    f"Optical engine failed materials signature check.  System signature is "
                    f"{sig} but needed {self.material_signature}")

        def check(fields, required, what):
            if bool(fields):
                sig = set(fields.keys())
                if not (sig >= required):
                    raise RuntimeError(
                        f"Optical engine failed {what} signature check.  System signature is "
                        f"{sig}, but needed {required}.")

        if self.dimension == 2:
            check(system._amalgamated_sources, SEGMENT_GEO_SIG | self.input_signature, "sources")
            for cat, sig in (("optical", self.optical_signature), ("stop", self.stop_signature),
                             ("target", self.target_signature)):
                check(getattr(system, f"_amalgamated_{cat}_segments"), SEGMENT_GEO_SIG | sig,
                      f"{cat} segments")
                check(getattr(system, f"_amalgamated_{cat}_arcs"), ARC_GEO_SIG | sig, f"{cat} arcs")
        else:
            check(system._amalgamated_sources, SOURCE_3D_SIG | self.input_signature, "sources")
            check(system._amalgamated_optical, TRIANGLE_GEO_SIG | self.optical_signature, "optical")
            check(system._amalgamated_stop, TRIANGLE_GEO_SIG | self.stop_signature, "stop")
            check(system._amalgamated_target, TRIANGLE_GEO_SIG | self.target_signature, "target")

    def validate_output(self):
        if self.dimension == 2:
            required = SEGMENT_GEO_SIG | self.output_signature
            for rays in (self.active_rays, self.finished_rays, self.stopped_rays, self.dead_rays):
                if bool(rays) and not (set(rays.keys()) >= required):
                    raise RuntimeError(
                        f"Optical engine failed output signature check.  System signature is "
                        f"{set(rays.keys())}, but needed {required}.")

    # ------------------------------------------------------------------------- history
    def _trace_mode(self, system=None):
        a = self.accelerate
        if a in (False, None, "all-pairs") or self._dimension != 3:
            return "all-pairs"
        if a in (True, "sort", "group"):
            return "group"
        fv = getattr(system if system is not None else self._optical_system,
                     "_merged_face_verts", None)
        return "group" if (fv is not None and fv.shape[0] >= 64) else "all-pairs"

    def clear_ray_history(self):
        self._history = {c: [] for c in _CLASSES}
        self._unfinished_rays = {}
        self._pending_trace = None

    # A fused optimiser step (fused_step.FusedStep) leaves the trace's outputs on the device and
    # never reads the ray counts; the ray sets are cut (one host read of the counts) when
    # somebody first asks for them.
    def _resolve_pending(self):
        pending = self._pending_trace
        if pending is not None:
            self._pending_trace = None
            self._history = {c: [] for c in _CLASSES}
            self._unfinished_rays = {}
            self._publish(pending())

    @property
    def last_trace(self):
        self._resolve_pending()
        return self._last_trace

    @last_trace.setter
    def last_trace(self, value):
        self._last_trace = value

    def _set(self, cls):
        self._resolve_pending()
        fields = amalgamate(self._history[cls])
        empty = None
        src = getattr(self, "_trace_src", None)
        if not fields and src is not None and bool(src):
            # (zero-length stand-ins of what the class would hold: see ReadOnlySet)
            geo = _GEO3 if self.dimension == 3 else _GEO2
            keys = set(geo) | (set(src.keys()) & self.simple_ray_inheritance)
            empty = {}
            for k in keys:
                try:
                    v = src[k]
                except KeyError:
                    continue
                empty[k] = v[:0]
        return ReadOnlySet(fields, empty)

    active_rays = property(lambda self: self._set("active"))
    finished_rays = property(lambda self: self._set("finished"))
    dead_rays = property(lambda self: self._set("dead"))
    stopped_rays = property(lambda self: self._set("stopped"))

    @property
    def unfinished_rays(self):
        self._resolve_pending()
        return self._unfinished_rays

    @property
    def all_rays(self):
        self._resolve_pending()
        h = self._history
        return ReadOnlySet(amalgamate(h["active"] + h["finished"] + h["dead"] + h["stopped"]))

    # --------------------------------------------------------------------------- trace
    def _flags(self):
        f = 0
        if self.compile_active_rays:
            f |= _lib.COMPILE_ACTIVE
        if self.compile_finished_rays:
            f |= _lib.COMPILE_FINISHED
        if self.compile_stopped_rays:
            f |= _lib.COMPILE_STOPPED
        if self.compile_dead_rays:
            f |= _lib.COMPILE_DEAD
        return f

    def _trace_inputs(self, rays, coherent_ok=True):
        """(ray block, scene arguments, merged face tensor) of a trace over the ray set ``rays``;
        ray block, n(lambda) table and scene arguments are cached per input tensor identity.
        With a coherent order (``coherent``, ``self._trace_perm`` is then the permutation) the
        block and the per-ray n(lambda) table come back PERMUTED: ray j is ``rays[perm[j]]``."""
        self._trace_perm = None
        system = self.optical_system
        geo = _GEO3 if self.dimension == 3 else _GEO2
        dt = self.ray_dtype or config.get_ray_dtype()
        index_mode, ghost = self._reaction()
        device_set = hasattr(rays, "ray_block")     # sources.DeviceRaySet: rays made in place
        if device_set:
            # (the buffers are persistent; ``cache_key`` changes with every update of the source;
            # the block in source order is only made when the trace runs in that order)
            key = rays.cache_key + (dt,)
            needs_grad = False
            block = None
        else:
            # the ray block and the n(lambda) table depend only on the input tensors: reuse them
            # while the caller hands in the very same tensors (static sources between steps).  The
            # cache holds the tensors, so their ids cannot be recycled under the key.
            needs_grad = any(rays[f].requires_grad for f in geo)
            key = tuple((id(rays[f]), rays[f]._version) for f in geo) + (dt,)
            cache = getattr(self, "_input_cache", None)
            if (cache is not None and cache[0] == key and not needs_grad
                    and all(a is rays[f] for a, f in zip(cache[2], geo))):
                block = cache[1]
            else:
                block = torch.stack([rays[f] for f in geo]).to(dt)
                self._input_cache = (key, block, [rays[f] for f in geo])
        n_table = None
        if index_mode:
            # keyed by the wavelengths' memory, not by the tensor object: a random source makes a
            # new (expanded) view of the same constant wavelength list every step.  The cache
            # holds the tensor, so its storage cannot be recycled under the key.
            wl = rays["wavelength"].detach()
            wkey = (wl.data_ptr(), tuple(wl.shape), wl.stride(), wl._version, wl.dtype,
                    tuple(id(m) for m in system.materials))
            tcache = getattr(self, "_table_cache", None)
            if tcache is not None and tcache[0] == wkey:
                n_table = tcache[1]
            else:
                # (one wavelength for every ray -- an expanded scalar --: one column, read by all)
                uniform = (self.dimension == 3 and wl.dim() == 1 and wl.shape[0] > 1
                           and wl.stride(0) == 0)
                n_table = system.material_table(wl[:1].contiguous() if uniform else wl)
                self._table_cache = (wkey, n_table, wl, uniform)
        mode = self._trace_mode(system)
        if self.dimension == 3:
            perm = None
            if coherent_ok and not needs_grad:
                perm = self._coherent_order(rays, block, n_table, key, mode, system, dt)
            if perm is not None:
                block, n_table = self._order_cache[2:4]
            elif block is None:
                block = rays.ray_block(dt)
            scene = system.scene_args(n_table, index_mode, ghost, cluster=mode != "all-pairs",
                                      deterministic=self.deterministic)
            tc = getattr(self, "_table_cache", None)
            scene.n_table_uniform = bool(index_mode and tc is not None and tc[3])
            scene.coherent_rays = perm is not None
            # a source that left no wavefront to the grouped kernel last time: no such launch
            ident = self._source_identity(rays, key)
            scene.coherent_only = (perm is not None
                                   and getattr(self, "_visit_all_key", None) == ident)
            scene.in_place = bool(scene.coherent_only and self.in_place is not False
                                  and not self.deterministic)
            self._visit_key = ident if perm is not None else None
            self._trace_perm = perm
        else:
            if block is None:
                block = rays.ray_block(dt)
            scene = system.scene_args(n_table, index_mode, ghost,
                                      finite_tir_gradient=self.finite_tir_gradient)
        fv = None
        if self.dimension == 3:
            fv = system._merged_face_verts
            if fv is None:
                fv = torch.zeros((0, 9), dtype=torch.float64, device=block.device)
        return block, scene, fv

    @staticmethod
    def _source_identity(rays, key):
        """What names a SOURCE across its updates: a device-made source keeps its identity when
        its rays are re-drawn (what the engine learns about its coherence stays valid), a set of
        plain tensors is its tensors."""
        return rays.identity if hasattr(rays, "identity") else key

    def _coherent_order(self, rays, block, n_table, key, mode, system, dt):
        """The coherent order of the source (see ``coherent``), with the permuted block and
        n(lambda) table: ``self._order_cache = (key, perm, block_p, n_table_p, n_table, held)``.
        A static source is ordered once (the cache holds its block and is honoured only for the
        very same one); a source re-drawn in place (``block`` is None: sources.DeviceRaySet) is
        ordered after every update, straight from its program, into the same persistent buffers."""
        device_set = block is None
        n = rays.n_rays if device_set else block.shape[1]
        on_gpu = rays.device.type == "cuda" if device_set else block.is_cuda
        if mode == "all-pairs" or self.coherent is False or n < 4096 or not on_gpu:
            return None
        ident = self._source_identity(rays, key)
        if self.coherent == "auto" and getattr(self, "_incoherent_key", None) == ident:
            return None                      # (tried: this source's wavefronts are no bundles)
        cached = getattr(self, "_order_cache", None)
        if cached is not None and cached[0] == key and (device_set or cached[5][0] is block):
            if cached[4] is not n_table:     # (other materials / wavelengths: same order)
                cached = cached[:3] + (self._permuted_table(n_table, cached[1]), n_table, cached[5])
                self._order_cache = cached
            return cached[1]
        fv = system._merged_face_verts
        fv = fv if fv is not None and fv.shape[0] else None
        if device_set:
            # (same source, new draw: the same perm buffer, so that a captured launch sequence of
            # the step stays valid)
            perm = None
            if cached is not None and cached[5][1] == ident and cached[1].numel() == n:
                perm = cached[1]
            # (re-made every step: the faster sort, ties in any order, unless the run must repeat
            # bit for bit -- the order is invisible in the results, only sums over rays round
            # differently)
            perm = rays.order(fv, out=perm, stable=bool(self.deterministic))
            block_p = rays.permuted(perm).ray_block(dt)
        else:
            perm = ops.ray_order(block, fv)
            block_p = ops.permute_rays(block, perm)
        self._order_cache = (key, perm, block_p, self._permuted_table(n_table, perm), n_table,
                             (block, ident))
        return perm

    def _order_inverse(self):
        """The inverse of the current trace's order (tfrt_scene3d.ray_slot), kept with the order it
        was made from (``_order_cache`` is replaced whenever the order is made again)."""
        perm = self._trace_perm
        if perm is None:
            return None
        oc = getattr(self, "_order_cache", None)
        kept = getattr(self, "_order_inv", None)
        if kept is None or kept[0] is not oc or kept[1] is not perm:
            kept = self._order_inv = (oc, perm, ops.inverse_order(perm))
        return kept[2]

    def _permuted_table(self, n_table, perm):
        if n_table is None:
            return None
        tc = getattr(self, "_table_cache", None)
        if tc is not None and tc[1] is n_table and tc[3]:
            return n_table                   # (one wavelength for every ray: the rows are constant)
        return ops.gather_rows(n_table, perm)

    def _note_left_over(self, left_over, passes=1):
        """Coherent-ray trace: remember whether the source left wavefronts to the grouped kernel
        (then the next trace of the same source launches it again) -- and, with coherent="auto",
        give the sorted trace up for this source when more than 5 % of its wavefront-passes were
        no narrow bundles (a light guide after a few bounces off its faceted wall: the cuts and the
        fallback then cost more than the shared walks save)."""
        key = getattr(self, "_visit_key", None)
        self._visit_all_key = key if (key is not None and int(left_over) == 0) else None
        if key is not None and self.coherent == "auto":
            n = getattr(self, "_order_cache", (None, None, None))[2]
            waves = max(1, (n.shape[1] // 64 if n is not None else 1) * max(int(passes), 1))
            if int(left_over) > 0.05 * waves:
                self._incoherent_key = key

    def _run(self, rays, max_passes, flags, predicted=None, lazy_ok=False):
        """One fused trace of ``max_passes`` passes over the ray set ``rays`` (field dict)."""
        # (tfrt_restore_order, which hands an ordered trace's ray sets back in the reference's
        # order, holds one scan chunk per RESTORE_CHUNK (pass, 32-ray word) pairs in 96 KB of LDS
        # and 1024 passes at most: a longer or larger trace runs in natural order, as it always did)
        n_rays = rays.n_rays if hasattr(rays, "n_rays") else rays[_GEO3[0]].shape[0]
        restorable = (max_passes <= 1024 and
                      ((n_rays + 31) // 32) * max(int(max_passes), 1) <= 24576 * _RESTORE_CHUNK)
        block, scene, fv = self._trace_inputs(rays, coherent_ok=restorable)
        if self.dimension == 3:
            # A source that is traced in place has shown that it leaves no wavefront over: nothing
            # of this trace needs to be known on the host before the next one is enqueued, so the
            # ray sets are cut (one host read of the counts) when somebody first asks for them --
            # the device works on this trace while the host prepares the next update()
            lazy = bool(lazy_ok and scene.in_place and predicted is None)
            out = ops.trace3d(block, fv, scene, max_passes, self.new_ray_length,
                              self.dead_ray_length, flags, predicted_counts=predicted,
                              perm=self._trace_perm, ray_slot=self._order_inverse(), lazy=lazy)
            if not lazy:
                self._note_left_over(out.get("left_over", 0), max_passes)
            return out
        return ops.trace2d(block, scene, max_passes, self.new_ray_length,
                           self.dead_ray_length, flags, predicted_counts=predicted)

    def _fields_from(self, out, cls, src, only_first_pass):
        """Field dict of one output class: geometry from the kernels, everything else gathered
        from the source set by source-ray index (simple inheritance, engine.py:2242-2281)."""
        geo = _GEO3 if self.dimension == 3 else _GEO2
        rays = out[cls]
        rows = out.get(cls + "_rows")
        carry = set(src.keys()) - set(geo)
        if not only_first_pass:
            carry &= self.simple_ray_inheritance
        memo = {}

        def ids():     # int64 source-ray indices, converted when a field is first gathered
            if "ids" not in memo:
                memo["ids"] = out[cls + "_id"].long()
            return memo["ids"]

        lazy = {f: (lambda f=f: src[f][ids()]) for f in carry}
        if rows is None:
            return LazyFields({g: rays[i] for i, g in enumerate(geo)}, lazy)
        # per-row autograd outputs, cut on first use: the gradient of a field stays one row
        geo_lazy = {g: (lambda i=i: rows[i]) for i, g in enumerate(geo)}
        geo_lazy.update(lazy)
        return LazyFields({}, geo_lazy)

    def ray_trace(self, max_iterations=25):
        """Trace the optical system (engine.py:2311-2330): all passes in one fused launch
        sequence on the device."""
        if not bool(self.optical_system):
            return
        self.clear_ray_history()
        src = self._source_set()
        if not src:
            return
        if self._custom_ops():
            # operations with their own main(): the reference's pass loop (engine.py:2311-2330),
            # one projection launch sequence + the operations' Python code per pass
            rays = {k: src[k] for k in src.keys()}
            for _ in range(int(max_iterations)):
                rays = self.single_pass(rays)
                if not bool(rays):
                    break
            self._unfinished_rays = rays if bool(rays) else {}
            return
        predicted = None
        sig = (src.n_rays if hasattr(src, "n_rays") else src["x_start"].shape[0],
               int(max_iterations), self._flags())
        if self.speculative_counts and self._predicted is not None and self._predicted[0] == sig:
            predicted = self._predicted[1]
        out = self._run(src, int(max_iterations), self._flags(), predicted,
                        lazy_ok=self.lazy_ray_sets and not self.speculative_counts)
        self._trace_sig, self._trace_src = sig, src
        if "finish" in out:
            self._pending_trace = out["finish"]       # (cut on first use: _resolve_pending)
            return
        self._publish(out)
        if "pending" not in out:
            self._predicted = (sig, out["raw_counts"])

    def _source_set(self):
        """The source rays this process traces: all of them, or this rank's contiguous block
        (``ray_shard``)."""
        src = self.optical_system._amalgamated_sources
        if not src:
            return src
        shard = self.ray_shard
        if shard == "auto":
            shard = (tdist.rank(), tdist.world_size()) if tdist.is_distributed() else None
        if shard is not None:
            if hasattr(src, "shard"):       # rays made in place: the rank makes its own block
                lo, hi = tdist.shard_bounds(src.n_rays, *shard)
                return src.shard(lo, hi)
            n = src["x_start"].shape[0]
            lo, hi = tdist.shard_bounds(n, *shard)
            # the same views for the same source tensors: everything cached per input tensor
            # (ray block, n(lambda) table) then also holds for a rank's shard of a static source
            skey = tuple((f, id(v), getattr(v, "_version", None)) for f, v in src.items()) + (lo, hi)
            cached = getattr(self, "_shard_cache", None)
            if cached is None or cached[0] != skey:
                cached = (skey, {f: v[lo:hi] for f, v in src.items()}, list(src.values()))
                self._shard_cache = cached
            src = cached[1]
        return src

    def verify_trace(self):
        """Resolve a speculative ray_trace(): returns True if the predicted counts were right;
        otherwise the ray sets have been rebuilt with the true counts (and the caller must
        re-evaluate whatever it derived from them)."""
        out = self._last_trace
        if out is None or "pending" not in out:
            return True
        ok, actual, fixed = out["pending"].resolve()
        self._predicted = (self._trace_sig, actual)
        if ok:
            out.pop("pending")
            return True
        self.clear_ray_history()
        fixed["raw_counts"] = actual
        self._publish(fixed)
        return False

    def _publish(self, out):
        src = self._trace_src
        self._last_trace = out
        counts = out["counts"]
        for k, cls in enumerate(_CLASSES):
            if cls not in out or out[cls].shape[1] == 0:
                continue
            later = bool(counts[1:, k].sum() > 0) if counts.shape[0] > 1 else False
            self._history[cls] = [self._fields_from(out, cls, src, not later)]
        geo = _GEO3 if self.dimension == 3 else _GEO2
        if out["unfinished"].shape[1]:
            ids = out["unfinished_id"].long()
            unf = {g: out["unfinished"][i] for i, g in enumerate(geo)}
            for f in self.simple_ray_inheritance & set(src.keys()):
                unf[f] = src[f][ids]
            self._unfinished_rays = unf

    def process_projection(self, input_rays):
        """One projection (engine.py:1544-2191): returns the reference's result dict and, like
        the reference, moves ``input_rays``' end points onto the boundaries they hit."""
        return self._single(input_rays)[0]

    def _single(self, input_rays):
        system = self.optical_system
        flags = self._flags() | _lib.COMPILE_ACTIVE
        out = self._run(input_rays, 1, flags)
        geo = _GEO3 if self.dimension == 3 else _GEO2
        result = {"rays": {}}
        for cls, on in (("active", True), ("finished", self.compile_finished_rays),
                        ("stopped", self.compile_stopped_rays), ("dead", self.compile_dead_rays)):
            if not on or cls not in out:
                continue
            fields = self._fields_from(out, cls, input_rays, True)
            result["rays"][cls] = fields
            if cls != "active" or self.compile_active_rays:
                self._history[cls].append(fields)
        # boundary data gathered to the reacting rays (engine.py:2113-2133)
        if self.dimension == 3:
            face = out["active_face"].long()
            optical = {f: v[face] for f, v in system._amalgamated_optical.items()
                       if f not in TRIANGLE_GEO_SIG} if bool(system._amalgamated_optical) else {}
            for f in TRIANGLE_GEO_SIG:
                if bool(system._merged):
                    optical[f] = system._merged[f][face]
            result["optical"] = optical
            if self.compile_technical_intersections:
                for cls, key, name, off in (
                        ("stopped", "stop", "_amalgamated_stop", system._optical_count),
                        ("finished", "target", "_amalgamated_target",
                         system._optical_count + system._stop_count)):
                    if cls in result["rays"] and bool(getattr(system, name)):
                        face = out[cls + "_face"].long()
                        d = {f: v[face - off] for f, v in getattr(system, name).items()}
                        d["norm"] = system._merged["norm"][face]
                        result[key] = d
        else:
            result["optical"] = ops.gather_optical_2d(system, out)
        # update the caller's ray dict with the projected end points
        n = input_rays[geo[0]].shape[0]
        half = len(geo) // 2
        for cls in ("active", "finished", "stopped"):
            if cls in out and out[cls].shape[1]:
                ids = out[cls + "_id"].long()
                for i in range(half, len(geo)):
                    col = input_rays[geo[i]].clone()
                    col[ids] = out[cls][i].to(col.dtype)
                    input_rays[geo[i]] = col
        new = {}
        if out["unfinished"].shape[1]:
            ids = out["unfinished_id"].long()
            new = {g: out["unfinished"][i] for i, g in enumerate(geo)}
            for f in self.simple_ray_inheritance:
                if f in input_rays:
                    new[f] = input_rays[f][ids]
        assert n == input_rays[geo[0]].shape[0]
        return result, new

    def single_pass(self, input_rays):
        """One pass (engine.py:2193-2302): project, react, inherit.  Returns the new ray set
        (``{}`` when no ray reacted)."""
        if not bool(self.optical_system):
            return {}
        result, new = self._single(input_rays)
        for cls in list(result["rays"].keys()):
            if result["rays"][cls]["x_start"].shape[0] == 0:
                result["rays"].pop(cls)
        self.last_projection_result = result
        for o in self._operations:
            o.preprocess(self, result)
        new_ray_dict = {}
        reaction = self._fused_op()
        if new and reaction is not None:
            valid = torch.ones(new["x_start"].shape[0], dtype=torch.bool, device=new["x_start"].device)
            new_ray_dict[reaction] = {"active": {"rays": new, "valid": valid}}
        # operations with their own main() (engine.py:2228-2234) + simple inheritance for the ray
        # sets they return (engine.py:2236-2281; the fused reaction's children already carry theirs)
        for o in self._custom_ops():
            op_result = o.main(self, result)
            if bool(op_result):
                for entry_type, entry in op_result.items():
                    source_set = result["rays"].get(entry_type)
                    if source_set is None:
                        continue
                    for sig in self.simple_ray_inheritance:
                        if sig in source_set:
                            entry["rays"][sig] = source_set[sig]
                new_ray_dict[o] = op_result
        for o in self._operations:
            o.postprocess(self, result, new_ray_dict)
        out_list = []
        for entry in new_ray_dict.values():
            for geo_entry in entry.values():
                v = geo_entry["valid"]
                out_list.append({k: f[v] for k, f in geo_entry["rays"].items()})
        return amalgamate(out_list)
