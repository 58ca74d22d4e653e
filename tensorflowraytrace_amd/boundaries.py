"""
Optical boundaries (tfrt/boundaries.py), host side.

A boundary is a dict-like set of per-primitive fields:

* segments : ``x_start, y_start, x_end, y_end``                       (SEGMENT_GEO_SIG)
* arcs     : ``x_center, y_center, angle_start, angle_end, radius``   (ARC_GEO_SIG)
* triangles: ``xp,yp,zp, x1,y1,z1, x2,y2,z2, norm``                   (TRIANGLE_GEO_SIG)

plus any number of extra per-primitive fields (``mat_in``, ``mat_out``, ``n_in`` ...).
Parametric boundaries hold a float64 ``Variable`` ``parameters`` (the thing an optimizer
moves); ``update()`` maps parameters -> vertices -> faces.  For triangles that last map and
its reverse run in HIP (``ops.build_faces`` -> tfrt_build_faces_forward/backward); triangle
geometry is kept as one (F,9) float64 tensor and the nine coordinate fields are column views
of it.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import config, ops
from . import mesh_tools as mt
from .update import RecursivelyUpdatable
from .variable import Variable

_TRI_COLS = {"xp": 0, "yp": 1, "zp": 2, "x1": 3, "y1": 4, "z1": 5, "x2": 6, "y2": 7, "z2": 8}


def amalgamate(stuff, signature=None):
    """Join field dicts by concatenation (engine.py:50-76).  Lives here as well as in
    ``engine`` because multi-boundaries need it and ``engine`` imports this module."""
    items = [s for s in stuff if bool(s)]
    if not items:
        return {}
    if not signature:
        signature = None
        for s in items:
            keys = set(s.keys())
            signature = keys if signature is None else (signature & keys)
    return {f: torch.cat([s[f] for s in items], 0) for f in signature}


_tap_log = None     # id(parameter) -> [aliases] while a collect_taps() block is open


class collect_taps:
    """``with collect_taps() as log: system.update()`` -- ``log[id(p)]`` lists the aliases through
    which the update read parameter ``p`` (a parameter may feed several boundaries).  Nothing is
    kept anywhere once the block is closed: no list grows with the number of updates and the
    parameter tensors stay plain leaves (``copy.deepcopy`` works)."""

    def __enter__(self):
        global _tap_log
        self._outer = _tap_log
        _tap_log = {}
        return _tap_log

    def __exit__(self, *exc):
        global _tap_log
        _tap_log = self._outer
        return False


def tap(parameters):
    """A fresh non-leaf alias of a parameter tensor, through which ``update()`` reads it.

    Differentiating the faces w.r.t. the alias gives the gradient w.r.t. the parameters without
    routing anything into the leaf's gradient accumulator.  That node lives as long as ANY old
    autograd graph of the parameter does and remembers the stream it was created on; a backward
    pass inside a HIP-graph capture (fused_step.FusedStep) that reaches an accumulator of another
    stream forks the capture onto that stream (the legacy stream, typically) and the runtime then
    crashes ending the capture.  The aliases are recorded only inside a ``collect_taps()`` block."""
    if not (isinstance(parameters, torch.Tensor) and parameters.requires_grad):
        return parameters
    alias = parameters.view_as(parameters)
    if _tap_log is not None:
        _tap_log.setdefault(id(parameters), []).append(alias)
    return alias


# ============================================================================ constraints

class Constraint(ABC):
    """Constraint between the parameters of two surfaces (boundaries.py:17-110).

    ``parent`` is ``"prev"`` (default; target/parent given to ``make`` are an index and the
    list of surfaces), ``"zero"``, ``"literal"`` or an integer index.
    """

    def __init__(self, parent="prev"):
        if type(parent) is int:
            if parent < 0:
                raise ValueError("Constraint: integer parent must be >= 0")
        elif type(parent) is str:
            if parent not in {"zero", "prev", "literal"}:
                raise ValueError("Constraint: string parent must be 'zero' or 'prev'.")
        self._parent = parent

    @property
    def parent(self):
        return self._parent

    def interpret_params(self, target, parent):
        if self._parent == "literal":
            return target.parameters, parent.parameters
        if self._parent == "zero":
            return target.parameters, torch.zeros_like(target.parameters)
        if self._parent == "prev":
            new_target = parent[target].parameters
            if target == 0:
                return new_target, torch.zeros_like(new_target)
            return new_target, parent[target - 1].parameters
        return parent[target].parameters, parent[self._parent].parameters

    @abstractmethod
    def make(self, target, parent):
        raise NotImplementedError


class NoConstraint(Constraint):
    def make(self, target, parent):
        return lambda: None


class PointConstraint(Constraint):
    """Fixes the parameter-space distance between two vertices (boundaries.py:124-158)."""

    def __init__(self, distance, target_vertex, parent_vertex=None, **kwargs):
        super().__init__(**kwargs)
        self.distance = distance
        self.target_vertex = target_vertex
        self.parent_vertex = parent_vertex or target_vertex

    def make(self, target, parent):
        def handler(target=target, parent=parent):
            t, p = self.interpret_params(target, parent)
            with torch.no_grad():
                diff = p[self.parent_vertex] - t[self.target_vertex] + self.distance
                t.add_(diff)
        return handler


class ThicknessConstraint(Constraint):
    """Fixes the min / max parameter-space distance between surfaces
    (boundaries.py:162-215): ``p += reduce(parent - p) + distance`` with reduce = max for
    mode 'min' and min for mode 'max'."""

    def __init__(self, distance, mode, **kwargs):
        super().__init__(**kwargs)
        self.distance = distance
        self.mode = mode

    @property
    def mode(self):
        return self._mode

    @mode.setter
    def mode(self, val):
        if val not in {"min", "max"}:
            raise ValueError("ThicknessConstraint: mode must be either 'min' or 'max'.")
        self._mode = val
        self._reduce = torch.max if val == "min" else torch.min

    def make(self, target, parent):
        def handler(target=target, parent=parent):
            t, p = self.interpret_params(target, parent)
            with torch.no_grad():
                t.add_(self._reduce(p - t) + self.distance)
        return handler


class ClipConstraint:
    """Clamps the parameters to [lower, upper] (boundaries.py:219-235)."""

    def __init__(self, lower, upper):
        self.lower = lower
        self.upper = upper

    def make(self, target, _):
        def handler(target=target):
            with torch.no_grad():
                target.parameters.clamp_(self.lower, self.upper)
        return handler


# ====================================================================== vector generators

class VectorGeneratorBase(ABC):
    """(N,3) zero points -> (N,3) unit vectors along which vertices move
    (boundaries.py:239-256)."""

    @abstractmethod
    def generate(self, zero):
        raise NotImplementedError

    @staticmethod
    def normalize(val):
        return val / torch.linalg.norm(val, dim=1, keepdim=True)


class SecondSurfaceVG(VectorGeneratorBase):
    def __init__(self, surface):
        self.surface = surface

    def generate(self, zero):
        return self.normalize(self.points.to(zero.device) - zero)

    @property
    def surface(self):
        return self._surface

    @surface.setter
    def surface(self, val):
        if type(val) is str:
            val = mt.read(val)
        if hasattr(val, "points"):
            self.points = val.points
            self._surface = val
        else:
            self.points = val
            self._surface = None

    @property
    def points(self):
        return self._points

    @points.setter
    def points(self, val):
        self._points = config.as_f64(val)


class FromPointVG(VectorGeneratorBase):
    def __init__(self, point):
        self.point = point

    def generate(self, zero):
        return self.normalize(zero - self.point.to(zero.device))

    @property
    def point(self):
        return self._point

    @point.setter
    def point(self, val):
        self._point = config.as_f64(val)


class FromVectorVG(VectorGeneratorBase):
    def __init__(self, vector):
        self.vector = vector

    def generate(self, zero):
        return self.normalize(self.vector.to(zero.device).expand(zero.shape).clone())

    @property
    def vector(self):
        return self._vector

    @vector.setter
    def vector(self, val):
        self._vector = config.as_f64(val)


class FromAxisVG(VectorGeneratorBase):
    """Vectors perpendicular to an axis, through the points (boundaries.py:353-383)."""

    def __init__(self, first, **kwargs):
        self.axis_point = config.as_f64(first)
        if "point" in kwargs:
            axis_vector = config.as_f64(kwargs["point"]) - self.axis_point
        elif "direction" in kwargs:
            axis_vector = config.as_f64(kwargs["direction"])
        else:
            raise ValueError(
                "FromAxisVG: Must provide a kwarg 'point' or 'direction' to define the axis.")
        self.axis_vector = self.normalize(axis_vector.reshape(1, 3))

    def generate(self, zero):
        ap = self.axis_point.to(zero.device)
        av = self.axis_vector.to(zero.device).expand(zero.shape)
        d = torch.sum((zero - ap) * av, dim=1)
        closest = ap + av * d.reshape(-1, 1)
        return self.normalize(zero - closest)


# =============================================================================== base

def _as_field(value, device):
    if isinstance(value, torch.Tensor):
        return value.to(device)
    arr = np.asarray(value)
    if arr.dtype.kind in "iub":
        return torch.as_tensor(arr, dtype=torch.int64, device=device)
    return torch.as_tensor(arr, dtype=torch.float64, device=device)


class BoundaryBase(RecursivelyUpdatable):
    """Dict-like boundary (boundaries.py:387-438).  ``material_dict`` entries become
    fields; scalars are broadcast to one value per primitive."""

    def __init__(self, name=None, material_dict={}, **kwargs):
        self._name = name
        if not hasattr(self, "_fields"):
            self._fields = {}
        self.material_dict = material_dict
        super().__init__(**kwargs)
        self.update_materials()

    @property
    def name(self):
        return self._name

    @property
    @abstractmethod
    def dimension(self):
        raise NotImplementedError

    @abstractmethod
    def update_materials(self):
        raise NotImplementedError

    def _update_materials(self, count, device):
        """Broadcast scalar material entries to one value per primitive (boundaries.py:423-428).
        The broadcast tensor is kept while value, count and device stay the same: update() runs
        every optimiser step, and a new tensor each time would look like a changed scene to the
        engine's caches (material columns, cluster order)."""
        cache = self.__dict__.setdefault("_material_cache", {})
        for field, raw in self.material_dict.items():
            scalar = None
            if not isinstance(raw, torch.Tensor) and np.ndim(raw) == 0:
                scalar = raw.item() if hasattr(raw, "item") else raw   # host value: no device read
            key = (scalar, type(scalar), int(count), str(device))
            hit = cache.get(field) if scalar is not None else None
            if hit is not None and hit[0] == key:
                self[field] = hit[1]
                continue
            value = _as_field(raw, device)
            if value.dim() < 1:
                value = value.expand(count).clone()
                if scalar is not None:
                    cache[field] = (key, value)
            self[field] = value

    def keys(self):
        return self._fields.keys()

    def __getitem__(self, key):
        return self._fields[key]

    def __setitem__(self, key, item):
        self._fields[key] = item

    def __bool__(self):
        return bool(self._fields)

    def __contains__(self, key):
        return key in self.keys()


# ================================================================================= 2-D

class _Manual2D(BoundaryBase):
    _shape_field = None

    @property
    def dimension(self):
        return 2

    def __setitem__(self, key, item):
        self._fields[key] = _as_field(item, config.get_device())

    def update_materials(self):
        if self._shape_field in self._fields:
            f = self._fields[self._shape_field]
            self._update_materials(f.shape[0], f.device)

    def _generate_update_handles(self):
        return []

    def _update(self):
        pass


class ArcBoundaryBase(_Manual2D):
    _shape_field = "x_center"


class ManualArcBoundary(ArcBoundaryBase):
    """Arcs given directly as fields (boundaries.py:458-473)."""


class SegmentBoundaryBase(_Manual2D):
    _shape_field = "x_start"


class ManualSegmentBoundary(SegmentBoundaryBase):
    """Segments given directly (boundaries.py:493-524)."""

    def feed_segments(self, segments):
        self["x_start"], self["y_start"], self["x_end"], self["y_end"] = \
            self.segment_splitter(segments)

    @staticmethod
    def segment_splitter(segments):
        s = config.as_f64(segments).reshape(-1, 4)
        return s[:, 0].clone(), s[:, 1].clone(), s[:, 2].clone(), s[:, 3].clone()


class ParametricSegmentBoundary(SegmentBoundaryBase):
    """Open polyline whose vertices slide between two matched point sets
    (boundaries.py:528-627): ``points = zero + p * (one - zero)``."""

    def __init__(self, zero_distribution, one_distribution, flip_norm=False,
                 initial_parameters=0.0, validate_shape=True, parameters=None, **kwargs):
        self._zero_distribution = zero_distribution
        self._one_distribution = one_distribution
        self.flip_norm = flip_norm
        if parameters is None:
            n = zero_distribution.points.shape[0]
            init = config.as_f64(initial_parameters).expand(n).clone()
            self.parameters = Variable(init)
        else:
            self.parameters = parameters
        self._fields = {}
        super().__init__(**kwargs)

    def __setitem__(self, key, item):
        self._fields[key] = item if isinstance(item, torch.Tensor) else _as_field(
            item, config.get_device())

    def _generate_update_handles(self):
        return [self._zero_distribution.update, self._one_distribution.update]

    def _update(self):
        self["x_start"], self["y_start"], self["x_end"], self["y_end"] = self._update_internal(
            self._zero_distribution.points, self._one_distribution.points, tap(self.parameters),
            self.flip_norm)

    @staticmethod
    def _update_internal(zero, one, parameter, flip_norm):
        zero = zero.to(parameter.device)
        one = one.to(parameter.device)
        points = zero + parameter.reshape(-1, 1) * (one - zero)
        if flip_norm:
            return points[1:, 0], points[1:, 1], points[:-1, 0], points[:-1, 1]
        return points[:-1, 0], points[:-1, 1], points[1:, 0], points[1:, 1]

    @property
    def zero_distribution(self):
        return self._zero_distribution

    @property
    def one_distribution(self):
        return self._one_distribution


def _listify(value, count, what, owner):
    try:
        if len(value) != count:
            raise ValueError(f"{owner}: constraints and {what} must have the same size.")
        return list(value)
    except TypeError:
        return [value] * count


class ParametricMultiSegmentBoundary(SegmentBoundaryBase):
    """Several ParametricSegmentBoundary layers over shared base points with constraints
    between them (boundaries.py:631-826)."""

    def __init__(self, zero_distribution, one_distribution, constraints, flip_norm,
                 initial_parameters=0.0, validate_shape=True, parameters=None,
                 material_list=[], **kwargs):
        owner = "ParametricMultiSegmentBoundary"
        try:
            self._surface_count = len(constraints)
        except TypeError as e:
            raise ValueError(f"{owner}: constraints must be iterable.") from e
        try:
            if len(flip_norm) != self._surface_count:
                raise ValueError(f"{owner}: constraints and flip_norm must have the same size.")
        except TypeError as e:
            raise ValueError(f"{owner}: flip_norm must be iterable.") from e
        self.flip_norm = list(flip_norm)
        if (isinstance(initial_parameters, (list, tuple))):
            initial_parameters = _listify(initial_parameters, self._surface_count,
                                          "initial_parameters", owner)
        else:
            initial_parameters = [initial_parameters] * self._surface_count
        if parameters is None:
            parameters = [None] * self._surface_count
        elif len(parameters) != self._surface_count:
            raise ValueError(f"{owner}: constraints and parameters must have the same size.")
        if len(material_list) == 0:
            material_list = [{}] * self._surface_count
        elif len(material_list) != self._surface_count:
            raise ValueError(f"{owner}: constraints and material_list must have the same size.")
        self._zero_distribution = zero_distribution
        self._one_distribution = one_distribution
        self.surfaces = [
            ParametricSegmentBoundary(zero_distribution, one_distribution, flip_norm=fn,
                                      initial_parameters=ip, parameters=p, material_dict=m,
                                      **kwargs)
            for fn, ip, p, m in zip(self.flip_norm, initial_parameters, parameters, material_list)
        ]
        self.constraints = constraints
        for i, (surface, constraint) in enumerate(zip(self.surfaces, constraints)):
            # (a ClipConstraint has no parent and takes the surface itself, like parent="zero")
            if getattr(constraint, "parent", "zero") != "zero":
                surface.update_handles.append(constraint.make(i, self.surfaces))
            else:
                surface.update_handles.append(constraint.make(surface, None))
        self._fields = {}
        super().__init__(**kwargs)

    def _update(self):
        self._fields = amalgamate(self.surfaces)

    def _generate_update_handles(self):
        return [self._zero_distribution.update, self._one_distribution.update] + \
            [s.update for s in self.surfaces]

    @property
    def surface_count(self):
        return self._surface_count

    @property
    def parameters(self):
        return [s.parameters for s in self.surfaces]


# ================================================================================= 3-D

class TriangleBoundaryBase(BoundaryBase):
    """Triangle mesh boundary (boundaries.py:830-938)."""

    def __init__(self, file_name=None, mesh=None, vertex_update_map=None, **kwargs):
        if file_name:
            self._mesh = mt.read(file_name)
        elif mesh is not None:
            self._mesh = mesh
        else:
            self._mesh = None
        self.vertex_update_map = vertex_update_map
        self._init_geometry()
        super().__init__(**kwargs)

    def _init_geometry(self):
        self._fields = {}
        if not hasattr(self, "_vertices"):
            self._vertices = None
        if not hasattr(self, "_faces"):
            self._faces = None
        self._face_verts = None
        self._norm = None
        self._faces_i32 = None
        self._mask_u8 = None

    @property
    def dimension(self):
        return 3

    # ``_face_verts`` / ``_norm``: the (F,9) / (F,3) tensors the kernels wrote.  A parametric
    # surface may have handed its update to the optical system's batch (ops.ParamFacesBatch: one
    # launch for all boundaries); whoever asks for the faces before the system has run it makes
    # it run what it has collected.
    def _faces_now(self):
        batch = self.__dict__.get("_faces_pending")
        if batch is not None:
            batch.flush()
            self.__dict__["_faces_pending"] = None

    @property
    def _face_verts(self):
        self._faces_now()
        return self.__dict__.get("_face_verts_value")

    @_face_verts.setter
    def _face_verts(self, value):
        self.__dict__["_face_verts_value"] = value

    @property
    def _norm(self):
        self._faces_now()
        return self.__dict__.get("_norm_value")

    @_norm.setter
    def _norm(self, value):
        self.__dict__["_norm_value"] = value

    # ---- dict protocol: geometry fields are column views of the (F,9) tensor
    def keys(self):
        ks = set(self._fields.keys())
        if self._face_verts is not None:
            ks |= set(_TRI_COLS) | {"norm"}
        return ks

    def __contains__(self, key):
        # (without building the key set: the system's merge asks this a dozen times per update)
        return key in self._fields or (
            (key in _TRI_COLS or key == "norm") and self._face_verts is not None)

    def __getitem__(self, key):
        if key in self._fields:
            return self._fields[key]
        if self._face_verts is not None:
            if key in _TRI_COLS:
                return self._face_verts[:, _TRI_COLS[key]]
            if key == "norm":
                return self._norm
        raise KeyError(key)

    def __setitem__(self, key, item):
        self._fields[key] = item if isinstance(item, torch.Tensor) else _as_field(
            item, config.get_device())

    def __bool__(self):
        return bool(self._fields) or self._face_verts is not None

    @property
    def face_verts(self):
        """(F,9) float64: xp,yp,zp,x1,y1,z1,x2,y2,z2 (what the trace kernels read)."""
        if self._face_verts is not None and not any(k in self._fields for k in _TRI_COLS):
            return self._face_verts
        return torch.stack([self[k] for k in _TRI_COLS], dim=1)

    def update_materials(self):
        fv = self._face_verts
        if fv is not None:
            self._update_materials(fv.shape[0], fv.device)
        elif "xp" in self._fields:
            self._update_materials(self._fields["xp"].shape[0], self._fields["xp"].device)

    @property
    def mesh(self):
        return self._mesh

    @property
    def vertices(self):
        return self._vertices

    @property
    def faces(self):
        return self._faces

    @property
    def vertex_update_map(self):
        return self._vertex_update_map

    @vertex_update_map.setter
    def vertex_update_map(self, val):
        self._vertex_update_map = None if val is None else np.asarray(val).astype(bool)
        self._mask_u8 = None

    def save(self, filename, **kwargs):
        if self._mesh is not None:
            self._mesh.save(filename, **kwargs)

    def update_vertices_from_mesh(self):
        if self._mesh is not None:
            self._vertices = config.as_f64(self._mesh.points)
            self._set_faces(self._mesh.triangles())

    def _set_faces(self, tri):
        tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        # keep the reference's (F,4) layout with the leading 3 for API compatibility
        self._faces = np.concatenate([np.full((tri.shape[0], 1), 3, dtype=np.int64), tri], axis=1)
        self._faces_i32 = None

    def _device_face_tables(self, dev):
        """(F,3) int32 vertex indices and the uint8 vertex_update_map (or None) on ``dev``."""
        if self._faces_i32 is None or self._faces_i32.device != dev:
            self._faces_i32 = torch.as_tensor(self._faces[:, 1:].astype(np.int32), device=dev)
        if self._vertex_update_map is not None and (
                self._mask_u8 is None or self._mask_u8.device != dev):
            self._mask_u8 = torch.as_tensor(
                self._vertex_update_map.astype(np.uint8), device=dev).contiguous()
        return self._faces_i32, (self._mask_u8 if self._vertex_update_map is not None else None)

    def update_fields_from_vertices(self):
        """vertices -> (F,9) face tensor + unit normals, in HIP (boundaries.py:890-923)."""
        if self._faces is None or self._vertices is None:
            return
        faces, mask = self._device_face_tables(self._vertices.device)
        self._face_verts, self._norm = ops.build_faces(self._vertices, faces, mask)
        for k in list(_TRI_COLS) + ["norm"]:
            self._fields.pop(k, None)

    def update_from_mesh(self):
        if self._mesh is not None:
            # a static mesh is uploaded and turned into faces once: update() runs every optimiser
            # step, and a host->device copy per step is slow and cannot sit in a captured launch
            # graph.  The host arrays are compared, so editing mesh.points in place is seen.
            pts = np.asarray(self._mesh.points)
            fcs = np.asarray(self._mesh.faces)
            seen = self.__dict__.get("_mesh_seen")
            if (seen is not None and self._face_verts is not None and seen[0].shape == pts.shape
                    and seen[1].shape == fcs.shape and np.array_equal(seen[0], pts)
                    and np.array_equal(seen[1], fcs)):
                return
            self.update_vertices_from_mesh()
            self.update_fields_from_vertices()
            self.__dict__["_mesh_seen"] = (pts.copy(), fcs.copy())

    def update_mesh_from_vertices(self):
        if self._vertices is not None and self._mesh is not None:
            self._mesh.points = self._vertices.detach().cpu().numpy()


class ManualTriangleBoundary(TriangleBoundaryBase):
    """Static mesh boundary (boundaries.py:942-963)."""

    def _generate_update_handles(self):
        return []

    def _update(self):
        self.update_from_mesh()


class ParametricTriangleBoundary(TriangleBoundaryBase):
    """Mesh whose vertices slide along per-vertex vectors: ``V = zero + p * vectors``
    (boundaries.py:967-1112).  ``flip_norm=True`` reverses every face (and the columns of
    the vertex_update_map) so the norm points the other way (boundaries.py:1022-1025,
    1096-1101)."""

    def __init__(self, zero_points, vector_generator, flip_norm=False, initial_parameters=0.0,
                 validate_shape=True, parameters=None, auto_update_mesh=False,
                 vertex_update_map=None, **kwargs):
        if type(zero_points) is str:
            zero_points = mt.read(zero_points)
        else:
            zero_points = zero_points.copy()
        if flip_norm:
            zero_points = self._flip_norm(zero_points)
            if vertex_update_map is not None:
                vertex_update_map = np.take(np.asarray(vertex_update_map), [2, 1, 0], axis=1)
        self._zero_points_mesh = zero_points
        self._zero_points = config.as_f64(zero_points.points)
        self.vertex_update_map = vertex_update_map
        self._mesh = mt.PolyData(zero_points.copy())
        self._init_geometry()
        TriangleBoundaryBase.update_vertices_from_mesh(self)

        self.vector_generator = vector_generator
        self.auto_update_mesh = auto_update_mesh
        self.reparametrize(self._zero_points)

        if parameters is None:
            init = config.as_f64(initial_parameters).expand(self._zero_points.shape[0]).clone()
            self.parameters = Variable(init)
        else:
            self.parameters = parameters
        BoundaryBase.__init__(self, **kwargs)
        if not self.auto_update_mesh:
            self.update_mesh_from_vertices()

    def _generate_update_handles(self):
        return []

    def _update(self):
        fused = (type(self)._update_internal is ParametricTriangleBoundary._update_internal
                 and self._faces is not None and self._zero_points.is_cuda
                 and isinstance(self.parameters, torch.Tensor) and self.parameters.is_cuda)
        if not fused:
            self._vertices = self._update_internal(self._zero_points, self._vectors,
                                                   tap(self.parameters))
            if self.auto_update_mesh:
                self.update_mesh_from_vertices()
            self.update_fields_from_vertices()
            return
        # one launch: parameters -> faces (ops.param_faces); the (V,3) vertex tensor is only
        # formed when somebody reads it (drawing, saving, a regulariser)
        self._vertices_pending = True
        # (drop the previous vertex tensor now: it would keep its autograd graph -- and the
        # parameters' gradient accumulator of whatever stream built it -- alive indefinitely)
        self.__dict__["_vertices_value"] = None
        faces, mask = self._device_face_tables(self._zero_points.device)
        batch = ops.current_faces_batch()
        if batch is not None:
            # (the optical system is updating: it runs the face updates of all its boundaries
            # in one launch when its handles are through)
            batch.add(self, tap(self.parameters), self._zero_points, self._vectors, faces, mask)
        else:
            self.__dict__["_faces_pending"] = None
            self._face_verts, self._norm = ops.param_faces(
                tap(self.parameters), self._zero_points, self._vectors, faces, mask)
        for k in list(_TRI_COLS) + ["norm"]:
            self._fields.pop(k, None)
        if self.auto_update_mesh:
            self.update_mesh_from_vertices()

    # ``_vertices`` of a parametric surface is computed on demand from the parameters when the
    # fused update ran (same expression as the two-step path, differentiable)
    @property
    def _vertices(self):
        d = self.__dict__
        if d.get("_vertices_pending"):
            d["_vertices_value"] = self._update_internal(
                self._zero_points, self._vectors, self.parameters)
            d["_vertices_pending"] = False
        return d.get("_vertices_value")

    @_vertices.setter
    def _vertices(self, value):
        self.__dict__["_vertices_value"] = value
        self.__dict__["_vertices_pending"] = False

    @staticmethod
    def _update_internal(zero, vectors, parameter):
        return zero + parameter.reshape(-1, 1) * vectors

    def reparametrize(self, zero_points):
        self._vectors = self.vector_generator.generate(self._zero_points)

    @property
    def zero_points(self):
        return self._zero_points_mesh

    @property
    def vectors(self):
        return self._vectors

    @staticmethod
    def _flip_norm(mesh):
        faces = np.reshape(mesh.faces, (-1, 4))
        faces = np.take(faces, [0, 3, 2, 1], axis=1)
        mesh.faces = np.reshape(faces, (-1,))
        return mesh

    def update_vertices_from_mesh(self):
        raise RuntimeError(
            "ParametricTriangleBoundary: update_vertices_from_mesh is disabled for parametric "
            "boundaries.")

    def update_from_mesh(self):
        raise RuntimeError(
            "ParametricTriangleBoundary: update_from_mesh is disabled for parametric boundaries.")


class MasterSlaveParametricTriangleBoundary(ParametricTriangleBoundary):
    """Fewer parameters than vertices: every slave vertex copies its master's parameter
    (boundaries.py:1116-1229).  The gather's reverse (scatter-add onto the masters) is
    torch's index backward."""

    def __init__(self, filter_masters, attach_slaves, *args, **kwargs):
        self._gather = None
        super().__init__(*args, **kwargs)
        verts = self._vertices.detach().cpu().numpy()
        masters = filter_masters(verts) if callable(filter_masters) else filter_masters
        masters = [int(m) for m in (np.nonzero(masters)[0] if np.asarray(masters).dtype == bool
                                    else masters)]
        master_index = {m: i for i, m in enumerate(masters)}
        unclaimed = set(range(verts.shape[0])) - set(masters)
        slave_masters = {}
        for master in masters:
            slaves = attach_slaves(verts, master, unclaimed)
            unclaimed -= set(slaves)
            for slave in slaves:
                slave_masters[slave] = master_index[master]
        gather = [master_index[i] if i in master_index else slave_masters[i]
                  for i in range(verts.shape[0])]
        self._gather = torch.as_tensor(gather, dtype=torch.int64, device=self._vertices.device)
        self.parameters = Variable(self.parameters.detach()[
            torch.as_tensor(masters, dtype=torch.int64, device=self._vertices.device)])
        self._update()
        self.update_materials()

    def _update(self):
        if self._gather is None:  # called from the parent constructor before the map exists
            return ParametricTriangleBoundary._update(self)
        params = tap(self.parameters)[self._gather].reshape(-1, 1)
        self._vertices = self._zero_points + params * self.vectors
        if self.auto_update_mesh:
            self.update_mesh_from_vertices()
        self.update_fields_from_vertices()


class ParametricMultiTriangleBoundary(TriangleBoundaryBase):
    """Several ParametricTriangleBoundary layers over one zero-point mesh with constraints
    between them (boundaries.py:1233-1412).  Feed ``.surfaces`` to ``system.optical``."""

    def __init__(self, zero_points, vector_generator, constraints, flip_norm,
                 initial_parameters=0.0, validate_shape=True, parameters=None,
                 material_list=[], **kwargs):
        owner = "ParametricMultiTriangleBoundary"
        if type(zero_points) is str:
            zero_points = mt.read(zero_points)
        self.zero_points = zero_points
        self.vector_generator = vector_generator
        try:
            self._surface_count = len(constraints)
        except TypeError as e:
            raise ValueError(f"{owner}: constraints must be iterable.") from e
        try:
            if len(flip_norm) != self._surface_count:
                raise ValueError(f"{owner}: constraints and flip_norm must have the same size.")
        except TypeError as e:
            raise ValueError(f"{owner}: flip_norm must be iterable.") from e
        if isinstance(initial_parameters, (list, tuple)):
            initial_parameters = _listify(initial_parameters, self._surface_count,
                                          "initial_parameters", owner)
        else:
            initial_parameters = [initial_parameters] * self._surface_count
        if parameters is None:
            parameters = [None] * self._surface_count
        elif len(parameters) != self._surface_count:
            raise ValueError(f"{owner}: constraints and parameters must have the same size.")
        if len(material_list) == 0:
            material_list = [{}] * self._surface_count
        elif len(material_list) != self._surface_count:
            raise ValueError(f"{owner}: constraints and material_list must have the same size.")

        self._surfaces = [
            ParametricTriangleBoundary(self.zero_points, self.vector_generator, flip_norm=fn,
                                       initial_parameters=ip, parameters=p, material_dict=m,
                                       **kwargs)
            for fn, ip, p, m in zip(flip_norm, initial_parameters, parameters, material_list)
        ]
        self.constraints = constraints
        for i, (surface, constraint) in enumerate(zip(self._surfaces, constraints)):
            if getattr(constraint, "parent", "zero") != "zero":
                surface.update_handles.append(constraint.make(i, self._surfaces))
            else:
                surface.update_handles.append(constraint.make(surface, None))
        self._mesh = None
        self.vertex_update_map = None
        self._init_geometry()
        RecursivelyUpdatable.__init__(self)

    def _update(self):
        self._fields = amalgamate(
            [{k: s[k] for k in s.keys()} for s in self._surfaces])

    def _generate_update_handles(self):
        return [s.update for s in self._surfaces]

    def update_materials(self):
        pass

    @property
    def surface_count(self):
        return self._surface_count

    @property
    def surfaces(self):
        return self._surfaces

    @property
    def parameters(self):
        return [s.parameters for s in self._surfaces]


class ParametricCylindricalGuide(TriangleBoundaryBase):
    """Closed cylinder-like light guide whose radius varies along (and optionally around) its
    axis (boundaries.py:1416-1717).

    ``parameters`` has ``z_res`` entries (``rotationally_symmetric=True``: one radius offset per
    ring) or ``z_res * theta_res`` entries (one per wall vertex); before every update the
    constraint ``p -= min(p)`` pins the thinnest point to ``minimum_radius``
    (boundaries.py:1613-1617).  Wall vertices move along ``FromAxisVG`` vectors; the optional
    cap-centre vertices stay on the axis.

    Difference from the reference: at HEAD the class drops the cap vertices from ``_vertices``
    but keeps face indices that still count them (boundaries.py:1607-1611, 1703-1717), which
    shifts every face by one vertex; here the vertex array keeps the cap vertices (fixed, zero
    displacement vector) so faces index what the mesh generator intended.
    """

    def __init__(self, start, end, minimum_radius, theta_res=6, z_res=8, start_cap=True,
                 end_cap=True, rotationally_symmetric=False, initial_parameters=0.0,
                 initial_taper=None, auto_update_mesh=False, use_vertex_update_map=True,
                 use_twist=False, **kwargs):
        self._mesh = mt.cylindrical_mesh(start, end, radius=minimum_radius, theta_res=theta_res,
                                         z_res=z_res, end_cap=end_cap, start_cap=start_cap,
                                         use_twist=use_twist)
        self._zero_points_mesh = self._mesh.copy()
        self._zero_points = config.as_f64(self._mesh.points)
        self._start_cap, self._end_cap = bool(start_cap), bool(end_cap)
        self._theta_res, self._z_res = int(theta_res), int(z_res)
        self._vertex_update_map = None
        self._init_geometry()
        self._vertices = self._zero_points
        self._set_faces(self._mesh.triangles())
        self._full_update_map, self._accumulator = mt.mesh_parametrization_tools(
            self._mesh, mt.get_closest_point(self._mesh, start))
        self.vector_generator = FromAxisVG(start, point=end)
        self.auto_update_mesh = auto_update_mesh
        self.reparametrize(self._zero_points)
        self.use_vertex_update_map = use_vertex_update_map
        self._rotationally_symmetric = bool(rotationally_symmetric)
        size = self._z_res if self._rotationally_symmetric else self._z_res * self._theta_res
        if initial_taper:
            try:
                t0, t1 = initial_taper[0], initial_taper[1]
            except (IndexError, TypeError) as e:
                raise ValueError(
                    "ParametricCylindricalGuide: initial_taper must be None or a 2-tuple.") from e
            init = torch.linspace(float(t0), float(t1), self._z_res, dtype=torch.float64,
                                  device=config.get_device())
            if not self._rotationally_symmetric:
                init = init.repeat_interleave(self._theta_res)
        else:
            init = config.as_f64(initial_parameters).expand(size).clone()
        self.parameters = Variable(init)
        BoundaryBase.__init__(self, **kwargs)
        if not self.auto_update_mesh:
            self.update_mesh_from_vertices()

    def _generate_update_handles(self):
        return []

    def _constraint(self):
        with torch.no_grad():
            self.parameters.sub_(self.parameters.min())

    def _update(self):
        self._constraint()
        p = tap(self.parameters)
        if self._rotationally_symmetric:
            p = p.repeat_interleave(self._theta_res)
        pads = []
        if self._start_cap:
            pads.append(torch.zeros(1, dtype=p.dtype, device=p.device))
        pads.append(p)
        if self._end_cap:
            pads.append(torch.zeros(1, dtype=p.dtype, device=p.device))
        p = torch.cat(pads) if len(pads) > 1 else p
        self._vertices = self._zero_points + p.reshape(-1, 1) * self._vectors
        if self.auto_update_mesh:
            self.update_mesh_from_vertices()
        self.update_fields_from_vertices()

    def reparametrize(self, zero_points):
        v = self.vector_generator.generate(self._zero_points)
        self._vectors = torch.where(torch.isfinite(v), v, torch.zeros_like(v))  # caps: on axis

    zero_points = property(lambda self: self._zero_points_mesh)
    vectors = property(lambda self: self._vectors)
    accumulator = property(lambda self: self._accumulator)
    rotationally_symmetric = property(lambda self: self._rotationally_symmetric)

    @property
    def use_vertex_update_map(self):
        return self._use_vertex_update_map

    @use_vertex_update_map.setter
    def use_vertex_update_map(self, val):
        self._use_vertex_update_map = bool(val)
        self.vertex_update_map = self._full_update_map if val else None

    def update_vertices_from_mesh(self):
        raise RuntimeError(
            "ParametricCylindricalGuide: update_vertices_from_mesh is disabled for parametric "
            "boundaries.")

    def update_from_mesh(self):
        raise RuntimeError(
            "ParametricCylindricalGuide: update_from_mesh is disabled for parametric boundaries.")
