"""
torch-facing wrappers of the HIP entry points (tensors are only device-array containers).

Everything here requires HIP tensors; CPU tensors raise ``TfrtError`` -- there is no CPU
fallback of the hot path.

* ``build_faces``            -> tfrt_build_faces_forward/backward  (autograd.Function)
* ``trace3d`` / ``Trace3D``  -> tfrt_trace3d_forward/backward      (autograd.Function)
* ``intersect3d``            -> tfrt_intersect3d                   (seam S1, engine.py:1103)
* ``snell3d`` / ``snell2d``  -> tfrt_snell3d / tfrt_snell2d        (seam S3, geometry.py:671/565)
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import TfrtError, RayOut, Scene3D, check

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64, torch.float16: _lib.F16}


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TfrtError(
                "tfrt kernels need tensors on a HIP device (got a CPU tensor); the hot path has "
                "no CPU fallback")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The current HIP stream of ``t``'s device as a ``void*`` (the raw accessor avoids building
    a ``torch.cuda.Stream`` object per C call: ~8 us each, eight calls per optimiser step)."""
    if _raw_stream is not None:
        idx = t.device.index
        return ctypes.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _c(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


# ------------------------------------------------------------------------- build_faces

_corner_cache = {}


def vertex_corners(faces, n_vertices):
    """(corner_start (V+1) i32, corner_list (3F) i32): the face corners ``f*3+c`` sorted by the
    vertex they reference, for the gather form of the reverse kernels (deterministic, no
    atomics).  Built once per face tensor (mesh topology) and cached with it."""
    key = (faces.data_ptr(), tuple(faces.shape), int(n_vertices), faces._version)
    hit = _corner_cache.get(key)
    if hit is not None and hit[0] is faces:
        return hit[1], hit[2]
    flat = faces.reshape(-1).long()
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=int(n_vertices))[:int(n_vertices)]
    start = torch.zeros(int(n_vertices) + 1, dtype=torch.int32, device=faces.device)
    start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    if len(_corner_cache) > 256:
        _corner_cache.clear()
    _corner_cache[key] = (faces, start.contiguous(), order.to(torch.int32).contiguous())
    return _corner_cache[key][1], _corner_cache[key][2]


class _BuildFaces(torch.autograd.Function):
    """vertices (V,3) f64 -> face_verts (F,9) f64, norm (F,3) f64.

    boundaries.py:890-923; ``update_mask`` (F,3) uint8 is the vertex_update_map
    (0 = stop_gradient for that corner, boundaries.py:900-913)."""

    @staticmethod
    def forward(ctx, vertices, faces, update_mask):
        _need_gpu(vertices, faces, update_mask)
        vertices = _c(vertices, torch.float64)
        F = faces.shape[0]
        fv = torch.empty((F, 9), dtype=torch.float64, device=vertices.device)
        norm = torch.empty((F, 3), dtype=torch.float64, device=vertices.device)
        check(_lib.lib().tfrt_build_faces_forward(
            _p(vertices), vertices.shape[0], _p(faces), F, _p(fv), _p(norm), _stream(vertices)),
            "tfrt_build_faces_forward")
        ctx.save_for_backward(fv, faces, update_mask)
        ctx.n_vertices = vertices.shape[0]
        if vertices.requires_grad and F > 0:
            ctx.corners = vertex_corners(faces, vertices.shape[0])   # (outside any graph capture)
        return fv, norm

    @staticmethod
    def backward(ctx, g_fv, g_norm):
        fv, faces, update_mask = ctx.saved_tensors
        g_fv = _c(g_fv, torch.float64)
        g_norm = _c(g_norm, torch.float64)
        if (g_fv is None and g_norm is None) or faces.shape[0] == 0:
            return torch.zeros((ctx.n_vertices, 3), dtype=torch.float64, device=fv.device), None, None
        start, lst = ctx.corners
        gv = torch.empty((ctx.n_vertices, 3), dtype=torch.float64, device=fv.device)
        check(_lib.lib().tfrt_build_faces_backward(
            _p(g_fv), _p(g_norm), _p(fv), _p(faces), _p(update_mask), faces.shape[0],
            ctx.n_vertices, _p(start), _p(lst), _p(gv), _stream(fv)), "tfrt_build_faces_backward")
        return gv, None, None


def build_faces(vertices, faces, update_mask=None):
    """faces: (F,3) int32 on the same device; update_mask: (F,3) uint8/bool or None."""
    faces = _c(faces, torch.int32)
    if update_mask is not None:
        update_mask = _c(update_mask, torch.uint8)
    return _BuildFaces.apply(vertices, faces, update_mask)


class _ParamFaces(torch.autograd.Function):
    """parameters (V,) -> face_verts (F,9), norm (F,3) of ``zero + parameters[:,None] * vectors``
    (boundaries.py:1065-1092 + 890-923) in one launch each way; see tfrt_param_faces_*."""

    @staticmethod
    def forward(ctx, parameters, zero_points, vectors, faces, update_mask):
        _need_gpu(parameters, zero_points, vectors, faces, update_mask)
        parameters = _c(parameters, torch.float64).reshape(-1)
        V, F = zero_points.shape[0], faces.shape[0]
        if parameters.shape[0] != V or vectors.shape != zero_points.shape:
            raise TfrtError("param_faces: parameters (V,), zero_points (V,3) and vectors (V,3) "
                            "must describe the same vertices")
        fv = torch.empty((F, 9), dtype=torch.float64, device=parameters.device)
        norm = torch.empty((F, 3), dtype=torch.float64, device=parameters.device)
        check(_lib.lib().tfrt_param_faces_forward(
            _p(zero_points), _p(vectors), _p(parameters), V, _p(faces), F, _p(fv), _p(norm),
            _stream(parameters)), "tfrt_param_faces_forward")
        ctx.save_for_backward(fv, faces, update_mask, vectors)
        ctx.shape = parameters.shape
        if F > 0:
            ctx.corners = vertex_corners(faces, V)
        ctx.set_materialize_grads(False)
        return fv, norm

    @staticmethod
    def backward(ctx, g_fv, g_norm):
        fv, faces, update_mask, vectors = ctx.saved_tensors
        g_fv = _c(g_fv, torch.float64)
        g_norm = _c(g_norm, torch.float64)
        if (g_fv is None and g_norm is None) or faces.shape[0] == 0:
            return torch.zeros(ctx.shape, dtype=torch.float64, device=fv.device), None, None, None, None
        start, lst = ctx.corners
        gp = torch.empty(ctx.shape, dtype=torch.float64, device=fv.device)
        check(_lib.lib().tfrt_param_faces_backward(
            _p(g_fv), _p(g_norm), _p(fv), _p(faces), _p(update_mask), _p(vectors),
            faces.shape[0], vectors.shape[0], _p(start), _p(lst), _p(gp), _stream(fv)),
            "tfrt_param_faces_backward")
        return gp, None, None, None, None


def param_faces(parameters, zero_points, vectors, faces, update_mask=None):
    """Faces of the parametric surface ``zero_points + parameters[:,None] * vectors``:
    (face_verts (F,9), norm (F,3)), differentiable w.r.t. ``parameters`` only (zero points and
    vectors are constants between reparametrisations, boundaries.py:1084-1085)."""
    faces = _c(faces, torch.int32)
    zero_points = _c(zero_points.detach(), torch.float64)
    vectors = _c(vectors.detach(), torch.float64)
    if update_mask is not None:
        update_mask = _c(update_mask, torch.uint8)
    return _ParamFaces.apply(parameters, zero_points, vectors, faces, update_mask)


class _ParamFacesMulti(torch.autograd.Function):
    """Several parametric surfaces (and the fixed boundaries between them) -> ONE merged
    face_verts block (M,9) and one norm block (sum of the parametric F,3), one launch each way
    (tfrt_param_faces_*_multi).  ``spec``: list of entries in the order of the merged block,
    ("param", zero_points, vectors, faces, update_mask) taking the next tensor of ``parameters``,
    or ("copy", face_verts (F,9) detached)."""

    @staticmethod
    def forward(ctx, spec, *parameters):
        dev = parameters[0].device if parameters else spec[0][1].device
        rows = [e[3].shape[0] if e[0] == "param" else e[1].shape[0] for e in spec]
        M = int(sum(rows))
        n_norm = int(sum(r for e, r in zip(spec, rows) if e[0] == "param"))
        fv = torch.empty((M, 9), dtype=torch.float64, device=dev)
        norm = torch.empty((n_norm, 3), dtype=torch.float64, device=dev)
        descs, keep, pars = [], [], []
        at = na = k = 0
        for e, F in zip(spec, rows):
            d = _lib.FaceSurface()
            d.n_faces = F
            d.face_verts = fv.data_ptr() + at * 72
            if e[0] == "param":
                _, zero, vectors, faces, mask = e
                par = _c(parameters[k], torch.float64).reshape(-1)
                k += 1
                V = zero.shape[0]
                if par.shape[0] != V or vectors.shape != zero.shape:
                    raise TfrtError("param_faces: parameters (V,), zero_points (V,3) and vectors "
                                    "(V,3) must describe the same vertices")
                d.zero_points, d.vectors, d.parameters = _p(zero), _p(vectors), _p(par)
                d.faces, d.n_vertices = _p(faces), V
                d.norm = norm.data_ptr() + na * 24
                pars.append((at, na, F, V, faces, mask, vectors, par.shape))
                keep.append(par)
                na += F
            else:
                d.copy_from = _p(e[1])
            descs.append(d)
            at += F
        stream = _stream(fv)
        for i in range(0, len(descs), _lib.MAX_SURFACES):
            chunk = descs[i:i + _lib.MAX_SURFACES]
            arr = (_lib.FaceSurface * len(chunk))(*chunk)
            check(_lib.lib().tfrt_param_faces_forward_multi(arr, len(chunk), stream),
                  "tfrt_param_faces_forward_multi")
        ctx.pars = pars
        ctx.corners = [vertex_corners(f, V) if F > 0 else None for (_, _, F, V, f, _, _, _) in pars]
        ctx.save_for_backward(fv)
        ctx.set_materialize_grads(False)
        return fv, norm

    @staticmethod
    def backward(ctx, g_fv, g_norm):
        (fv,) = ctx.saved_tensors
        g_fv = _c(g_fv, torch.float64)
        g_norm = _c(g_norm, torch.float64)
        out, descs = [], []
        for (at, na, F, V, faces, mask, vectors, shape), corners in zip(ctx.pars, ctx.corners):
            if (g_fv is None and g_norm is None) or F == 0:
                out.append(torch.zeros(shape, dtype=torch.float64, device=fv.device))
                continue
            sink = GradSink._active
            gp = sink.take(shape) if sink is not None else None
            if gp is None:
                gp = torch.empty(shape, dtype=torch.float64, device=fv.device)
            d = _lib.FaceSurfaceGrad()
            d.grad_face_verts = g_fv.data_ptr() + at * 72 if g_fv is not None else None
            d.grad_norm = g_norm.data_ptr() + na * 24 if g_norm is not None else None
            d.face_verts = fv.data_ptr() + at * 72
            d.update_mask, d.vectors = _p(mask), _p(vectors)
            d.corner_start, d.corner_list = _p(corners[0]), _p(corners[1])
            d.n_vertices, d.grad_parameters = V, _p(gp)
            descs.append(d)
            out.append(gp)
        stream = _stream(fv)
        for i in range(0, len(descs), _lib.MAX_SURFACES):
            chunk = descs[i:i + _lib.MAX_SURFACES]
            arr = (_lib.FaceSurfaceGrad * len(chunk))(*chunk)
            check(_lib.lib().tfrt_param_faces_backward_multi(arr, len(chunk), stream),
                  "tfrt_param_faces_backward_multi")
        return (None, *out)


_faces_batch = None   # the batch parametric boundaries hand their update to (see ParamFacesBatch)


class GradSink:
    """While active (``with GradSink(flat):``) the parameter gradients of the face updates are
    written into consecutive slices of ``flat`` (a persistent float64 buffer) instead of fresh
    tensors: a sharded optimiser step then all-reduces ``flat`` as it is -- no concatenation
    before the collective, no slicing after it."""

    _active = None

    def __init__(self, flat):
        self.flat, self.at, self.taken = flat, 0, []

    def __enter__(self):
        self._prev, GradSink._active = GradSink._active, self
        return self

    def __exit__(self, *exc):
        GradSink._active = self._prev
        return False

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        if self.at + n > self.flat.numel():
            return None
        view = self.flat[self.at:self.at + n].view(shape)
        self.at += n
        self.taken.append(view)
        return view

    def holds(self, t):
        """True if ``t`` is one of the slices handed out (same memory, same size)."""
        return any(v.data_ptr() == t.data_ptr() and v.numel() == t.numel() for v in self.taken)


def current_faces_batch():
    return _faces_batch


class ParamFacesBatch:
    """The face updates of one optical system's update(), collected and run as one launch.

    ``with batch:`` -- inside, ParametricTriangleBoundary._update registers its parameters here
    (``add``) instead of launching; ``flush(order)`` runs everything in ONE launch and hands every
    boundary its rows of the merged block as ``_face_verts`` / ``_norm``.  ``order``: the
    boundaries of the system in the order of its merged face block; boundaries without a request
    whose faces are fixed (no gradient) are copied into place, so the block IS the system's merged
    face tensor (``merged``) and nothing is concatenated afterwards.  A boundary that is asked
    for its faces before that flushes what has been collected so far."""

    def __init__(self):
        self.requests = []      # (boundary, parameters (tap), zero_points, vectors, faces, mask)
        self.merged = None      # (face block, [(boundary, first row, end row)]) after a full flush
        self._outer = None

    def __enter__(self):
        global _faces_batch
        self._outer, _faces_batch = _faces_batch, self
        return self

    def __exit__(self, *exc):
        global _faces_batch
        _faces_batch = self._outer
        return False

    def add(self, boundary, parameters, zero_points, vectors, faces, mask):
        faces = _c(faces, torch.int32)
        zero_points = _c(zero_points.detach(), torch.float64)
        vectors = _c(vectors.detach(), torch.float64)
        if mask is not None:
            mask = _c(mask, torch.uint8)
        _need_gpu(parameters, zero_points, vectors, faces, mask)
        self.requests.append((boundary, parameters, zero_points, vectors, faces, mask))
        boundary.__dict__["_faces_pending"] = self

    def flush(self, order=None):
        if not self.requests:
            return
        reqs, self.requests = self.requests, []
        by_boundary = {id(r[0]): r for r in reqs}
        layout = None
        if order is not None:
            layout = []
            for b in order:
                r = by_boundary.get(id(b))
                if r is not None:
                    layout.append(("param", r))
                    continue
                fv = b.__dict__.get("_face_verts_value")
                # (a boundary whose xp .. z2 fields were assigned by hand no longer shows its raw
                # vertex block: its face_verts property stacks the fields -- not copyable)
                overridden = any(k in getattr(b, "_fields", {}) for k in
                                 ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2"))
                if (fv is None or overridden or fv.requires_grad or not fv.is_cuda
                        or fv.dtype != torch.float64
                        or not fv.is_contiguous() or fv.dim() != 2 or fv.shape[1] != 9):
                    layout = None   # (something the block cannot hold: faces only, then cat)
                    break
                layout.append(("copy", b, fv))
            if layout is not None and sum(1 for e in layout if e[0] == "param") != len(reqs):
                layout = None
        if layout is None:
            layout = [("param", r) for r in reqs]
        spec = [("param", e[1][2], e[1][3], e[1][4], e[1][5]) if e[0] == "param"
                else ("copy", e[2]) for e in layout]
        pars = [e[1][1] for e in layout if e[0] == "param"]
        fv, norm = _ParamFacesMulti.apply(spec, *pars)
        at = na = 0
        rows = []
        for e in layout:
            if e[0] == "param":
                b, F = e[1][0], e[1][4].shape[0]
                b.__dict__["_faces_pending"] = None
                b._face_verts = fv[at:at + F]
                b._norm = norm[na:na + F]
                rows.append((b, at, at + F))
                na += F
            else:
                F = e[2].shape[0]
                rows.append((e[1], at, at + F))
            at += F
        if order is not None and len(layout) == len(order):
            self.merged = (fv, rows)


# ------------------------------------------------------------------- pairwise geometry

def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return ctypes.cast(arr, ctypes.c_void_p), arr


def _pairwise(first, second, grid):
    """Common operand handling of the dense geometry functions: returns (n_cols, n_rows,
    first tensors, first strides, second tensors, second strides, output shape).  ``grid``:
    meshgrid form (first set along columns, second along rows); otherwise element-wise."""
    first = [_c(torch.as_tensor(t), torch.float64) for t in first]
    dev = first[0].device
    second = [_c(torch.as_tensor(t, device=dev), torch.float64) for t in second]
    _need_gpu(*first, *second)
    if grid:
        first = [t.reshape(-1) for t in first]
        second = [t.reshape(-1) for t in second]
        n_cols, n_rows = first[0].numel(), second[0].numel()
        if any(t.numel() != n_cols for t in first) or any(t.numel() != n_rows for t in second):
            raise TfrtError("pairwise geometry: operands of one set must have equal lengths")
        return n_cols, n_rows, first, (1, 0), second, (0, 1), (n_rows, n_cols)
    shape = torch.broadcast_shapes(*[t.shape for t in first + second])
    first = [t.expand(shape).contiguous().reshape(-1) for t in first]
    second = [t.expand(shape).contiguous().reshape(-1) for t in second]
    return first[0].numel(), 1, first, (1, 0), second, (1, 0), tuple(shape)


def line_intersect(first, second, epsilion, grid):
    """geometry.py:27-167.  first = (x1s, y1s, x1e, y1e), second = (x2s, y2s, x2e, y2e).
    Returns x, y, valid, u, v."""
    nc, nr, f, fs, s2, ss, shape = _pairwise(first, second, grid)
    dev = f[0].device
    new = lambda: torch.empty(shape, dtype=torch.float64, device=dev)
    x, y, u, v = new(), new(), new(), new()
    valid = torch.empty(shape, dtype=torch.uint8, device=dev)
    fp, keep1 = _ptr_array(f)
    sp, keep2 = _ptr_array(s2)
    check(_lib.lib().tfrt_line_intersect(nc, nr, fp, fs[0], fs[1], sp, ss[0], ss[1],
                                         float(epsilion), _p(x), _p(y), _p(valid), _p(u), _p(v),
                                         _stream(x)), "tfrt_line_intersect")
    return x, y, valid.bool(), u, v


def line_triangle_intersect(rays, triangles, epsilion, grid):
    """geometry.py:191-320.  rays = (rx1..rz2), triangles = (xp..z2).
    Returns x, y, z, valid, ray_u, trig_u, trig_v."""
    nc, nr, f, fs, s2, ss, shape = _pairwise(rays, triangles, grid)
    dev = f[0].device
    new = lambda: torch.empty(shape, dtype=torch.float64, device=dev)
    x, y, z, ru, tu, tv = new(), new(), new(), new(), new(), new()
    valid = torch.empty(shape, dtype=torch.uint8, device=dev)
    fp, keep1 = _ptr_array(f)
    sp, keep2 = _ptr_array(s2)
    check(_lib.lib().tfrt_line_triangle_intersect(
        nc, nr, fp, fs[0], fs[1], sp, ss[0], ss[1], float(epsilion), _p(x), _p(y), _p(z),
        _p(valid), _p(ru), _p(tu), _p(tv), _stream(x)), "tfrt_line_triangle_intersect")
    return x, y, z, valid.bool(), ru, tu, tv


def line_circle_intersect(lines, circles, epsilion, grid):
    """geometry.py:338-547.  lines = (xs, ys, xe, ye), circles = (xc, yc, r).  Returns the
    (plus, minus) dicts with x, y, valid, u, v."""
    nc, nr, f, fs, s2, ss, shape = _pairwise(lines, circles, grid)
    dev = f[0].device
    new = lambda: torch.empty(shape, dtype=torch.float64, device=dev)
    roots = []
    for _ in range(2):
        roots.append(([new(), new(), new(), new()], torch.empty(shape, dtype=torch.uint8, device=dev)))
    fp, keep1 = _ptr_array(f)
    sp, keep2 = _ptr_array(s2)
    pp, keep3 = _ptr_array(roots[0][0])
    mp, keep4 = _ptr_array(roots[1][0])
    check(_lib.lib().tfrt_line_circle_intersect(
        nc, nr, fp, fs[0], fs[1], sp, ss[0], ss[1], float(epsilion), pp, _p(roots[0][1]), mp,
        _p(roots[1][1]), _stream(roots[0][1])), "tfrt_line_circle_intersect")
    out = []
    for (x, y, u, v), valid in roots:
        out.append({"x": x, "y": y, "valid": valid.bool(), "u": u, "v": v})
    return out[0], out[1]


# ------------------------------------------------------------------- parameter update

def sgd_process(grad, scale, clip, param=None, sgd_learning_rate=0.0):
    """optimizer.py:223-247 (+ :316 when ``param`` is given) in one launch: returns
    ``clip(where(isfinite(grad), grad, 0) * scale, -clip, clip)`` and, if ``param`` is given,
    also applies ``param -= sgd_learning_rate * processed`` in place."""
    _need_gpu(grad, param)
    if grad.dtype not in (torch.float32, torch.float64):
        raise TfrtError(f"sgd_process: float32/float64 gradients only, got {grad.dtype}")
    grad = grad.contiguous()
    out = torch.empty_like(grad)
    if param is not None and (param.dtype != grad.dtype or param.shape != grad.shape
                              or not param.is_contiguous()):
        raise TfrtError("sgd_process: param must be contiguous with the gradient's shape and dtype")
    check(_lib.lib().tfrt_sgd_process(
        _p(grad), _p(out), _p(param), grad.numel(), _DT[grad.dtype], float(scale), float(clip),
        float(sgd_learning_rate), _stream(grad)), "tfrt_sgd_process")
    return out


class CsrMatrix:
    """A square accumulator / smoother matrix held in CSR form on the device
    (optimizer.py:250-255, 277-282 multiply dense (P,P) matrices whose rows hold a few
    non-zeros, mesh_tools.py:221-421)."""

    def __init__(self, matrix, device):
        if isinstance(matrix, torch.Tensor):
            dense = matrix.detach().to_dense() if matrix.layout != torch.strided else matrix.detach()
            dense = dense.to(device="cpu", dtype=torch.float64).numpy()
        else:
            dense = np.asarray(matrix, dtype=np.float64)
        if dense.ndim != 2:
            raise TfrtError("CsrMatrix: need a rank-2 matrix")
        rows, cols = np.nonzero(dense)                      # row-major order = CSR order
        crow = np.zeros(dense.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=dense.shape[0]), out=crow[1:])
        self.shape = tuple(dense.shape)
        self.nnz = int(rows.size)
        self.crow = torch.as_tensor(crow).to(device)
        self.col = torch.as_tensor(cols.astype(np.int64)).to(device)
        self.val = torch.as_tensor(np.ascontiguousarray(dense[rows, cols])).to(device)

    def matvec(self, x):
        """A @ x for a (P,) or (P,1) float64 vector on the matrix's device."""
        _need_gpu(x)
        shape = x.shape
        v = _c(x.reshape(-1), torch.float64)
        if v.shape[0] != self.shape[1]:
            raise TfrtError(f"CsrMatrix.matvec: matrix is {self.shape}, vector has {v.shape[0]}")
        y = torch.empty(self.shape[0], dtype=torch.float64, device=v.device)
        check(_lib.lib().tfrt_csr_matvec(_p(self.crow), _p(self.col), _p(self.val), _p(v), _p(y),
                                         self.shape[0], _stream(v)), "tfrt_csr_matvec")
        return y.reshape(shape).to(x.dtype) if self.shape[0] == self.shape[1] else y


# ----------------------------------------------------------------------------- trace3d

class Scene3DArgs:
    """Device tensors + scalars describing the merged 3-D boundary set (tfrt_scene3d)."""

    def __init__(self, face_verts, catagory, mat_in=None, mat_out=None, n_in=None, n_out=None,
                 n_table=None, intersect_epsilion=1e-10, size_epsilion=1e-10,
                 ray_start_epsilion=1e-10, face_grad_mask=None, cluster_order=None,
                 deterministic=False, coherent_rays=False):
        self.face_verts = face_verts  # (M,9) f64, may require grad
        self.catagory = _c(catagory, torch.int32)
        self.mat_in = _c(mat_in, torch.int32)
        self.mat_out = _c(mat_out, torch.int32)
        self.n_in = _c(n_in, torch.float64)
        self.n_out = _c(n_out, torch.float64)
        # "value" mode: the tensors as handed in (they may require grad: d error / d index)
        self.n_in_arg = n_in if isinstance(n_in, torch.Tensor) else None
        self.n_out_arg = n_out if isinstance(n_out, torch.Tensor) else None
        self.n_table = _c(n_table, torch.float64)  # (n_materials, N)
        # True: `n_table` has ONE column that holds for every ray (one wavelength)
        self.n_table_uniform = False
        self.face_grad_mask = _c(face_grad_mask, torch.uint8)  # (M) or None
        self.cluster_order = _c(cluster_order, torch.int32)    # (M) or None: two-level filter
        self.deterministic = bool(deterministic)               # ordered reverse-sweep sums
        # True: the rays handed to the trace are in a coherent order (see ray_order()): wavefronts
        # share one walk of the hierarchy (k_intersect_beam), the reverse sweep sums per wavefront.
        # May be reassigned between traces (it belongs to the source, not to the boundaries).
        self.coherent_rays = bool(coherent_rays)
        # with coherent_rays: True = launch no grouped kernel behind k_intersect_beam (a source
        # whose earlier traces left no wavefront over: out["left_over"] == 0)
        self.coherent_only = False
        # with coherent_rays: all passes in ONE launch, rays kept in place (tfrt_scene3d.in_place)
        self.in_place = False
        self.eps = (float(intersect_epsilion), float(size_epsilion), float(ray_start_epsilion))

    def struct(self, face_verts):
        M = face_verts.shape[0]
        cached = getattr(self, "_struct_cache", None)
        if cached is not None and cached[0] == M:
            # every other field points at tensors this object owns: only the face pointer moves
            sc = cached[1]
            sc.face_verts = face_verts.data_ptr() if M else None
            sc.coherent_rays = 1 if self.coherent_rays else 0
            sc.coherent_only = 1 if (self.coherent_only and self.coherent_rays) else 0
            sc.n_table_uniform = 1 if self.n_table_uniform else 0
            sc.in_place = 1 if (self.in_place and self.coherent_rays) else 0
            return sc
        sc = Scene3D()
        sc.face_verts = face_verts.data_ptr() if M else None
        sc.catagory = self.catagory.data_ptr() if M else None
        for name in ("mat_in", "mat_out", "n_in", "n_out"):
            t = getattr(self, name)
            setattr(sc, name, t.data_ptr() if (t is not None and M) else None)
        sc.n_faces = M
        if self.n_table is not None and self.n_table.numel():
            sc.n_table = self.n_table.data_ptr()
            sc.n_table_stride = self.n_table.shape[1]
            sc.n_materials = self.n_table.shape[0]
        else:
            sc.n_table, sc.n_table_stride, sc.n_materials = None, 0, 0
        sc.intersect_epsilion, sc.size_epsilion, sc.ray_start_epsilion = self.eps
        g = self.face_grad_mask
        sc.face_grad_mask = g.data_ptr() if (g is not None and M) else None
        co = self.cluster_order
        if co is not None and co.numel() != M:
            raise TfrtError("cluster_order must be a permutation of the M face indices")
        sc.cluster_order = co.data_ptr() if (co is not None and M) else None
        sc.n_table_uniform = 1 if self.n_table_uniform else 0
        sc.deterministic = 1 if self.deterministic else 0
        sc.coherent_rays = 1 if self.coherent_rays else 0
        sc.coherent_only = 1 if (self.coherent_only and self.coherent_rays) else 0
        sc.in_place = 1 if (self.in_place and self.coherent_rays) else 0
        self._struct_cache = (M, sc)
        return sc


class TraceTape:
    """Everything the reverse sweep needs (kept alive by the autograd node)."""
    pass


def _ray_out(rays, ids, faces):
    o = RayOut()
    if rays is None:
        o.rays, o.ray_id, o.face, o.capacity = None, None, None, 0
    else:
        o.rays, o.ray_id, o.face, o.capacity = rays.data_ptr(), ids.data_ptr(), faces.data_ptr(), rays.shape[1]
    return o


class _IntPool:
    """The int32 outputs of one trace (counts + per-class source ids and faces) cut from one
    allocation: one fill instead of one per array when they must start as zeros."""

    def __init__(self, dev, zero_all, n_counts, flags, cap_n, passes):
        caps = ((_lib.COMPILE_FINISHED, cap_n), (_lib.COMPILE_ACTIVE, cap_n * max(passes, 1)),
                (_lib.COMPILE_STOPPED, cap_n), (_lib.COMPILE_DEAD, cap_n))
        pad = self._pad
        total = pad(n_counts) + pad(cap_n) + sum(2 * pad(c) for f, c in caps if flags & f)
        if zero_all:
            self._buf = torch.zeros(total, dtype=torch.int32, device=dev)
        else:
            self._buf = torch.empty(total, dtype=torch.int32, device=dev)
            self._buf[:n_counts].zero_()
        self.counts = self._buf[:n_counts]
        self._at = pad(n_counts)

    @staticmethod
    def _pad(n):          # every array starts on a 256-byte boundary, like its own allocation
        return (n + 63) & ~63

    def take(self, n):
        out = self._buf[self._at:self._at + n]
        self._at += self._pad(n)
        return out


def _with_rows(blocks, present):
    """Outputs of a trace Function: the class blocks followed by the rows of every compiled class
    (views of the same memory).  A consumer that differentiates single fields of a class (``y_end``
    of the finished rays, say) takes the row outputs: autograd then hands backward() that row's
    gradient alone instead of assembling a dense zero-padded block per field it touched
    (select_backward + slice_backward + add over (rows, capacity): ~90 us per step at 1M rays)."""
    rows = []
    for b, p in zip(blocks, present):
        if p:
            rows.extend(b.unbind(0))
    return tuple(blocks) + tuple(rows)


def _split_rows(outs, present, n_rows):
    """(class blocks, {class name: its row outputs}) from the outputs of a trace Function."""
    rows, at = {}, 4
    for name, p in zip(_CLASS_NAMES, present):
        if p:
            rows[name] = outs[at:at + n_rows]
            at += n_rows
    return outs[:4], rows


def _class_grads(ctx_present, caps, n_rows, dev, grads):
    """Float64 (rows, capacity) gradient block per class from the block / row gradients that
    autograd delivered (None where the class took no gradient at all)."""
    blocks, rows = grads[:4], grads[4:]
    out, at = [], 0
    for k, present in enumerate(ctx_present):
        if not present:
            out.append(None)
            continue
        gb, gr = blocks[k], rows[at:at + n_rows]
        at += n_rows
        if gb is None and all(g is None for g in gr):
            out.append(None)
            continue
        if all(g is None for g in gr):
            out.append(_c(gb, torch.float64))
            continue
        if gb is None:
            blk = torch.zeros((n_rows, caps[k]), dtype=torch.float64, device=dev)
        else:
            blk = gb.to(torch.float64, copy=True).contiguous()
        for i, g in enumerate(gr):
            if g is not None:
                if gb is None:
                    blk[i].copy_(g)
                else:
                    blk[i].add_(g)
        out.append(blk)
    return out


class _Trace3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, face_verts, n_in, n_out, scene, opts):
        # (n_in / n_out: the per-face indices of "value" mode when they require grad -- the values
        # are read through `scene`; listing them here gives their gradients a place to go)
        _need_gpu(src, face_verts)
        dev = src.device
        if src.dtype not in _DT:
            raise TfrtError(f"ray state dtype must be float32, float64 or float16, got {src.dtype}")
        src = src.contiguous()
        face_verts = _c(face_verts, torch.float64)
        N = src.shape[1]
        P = int(opts["max_passes"])
        flags = int(opts["flags"])
        dt = _DT[src.dtype]
        L = _lib.lib()
        wsb = L.tfrt_trace3d_workspace_bytes(N, face_verts.shape[0], P, dt)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        # With speculative slicing (predicted counts) rows beyond the true counts may be read
        # before the prediction is verified: the index arrays must then hold valid indices
        # (zeros); the ray blocks may hold anything, a wrong guess is re-evaluated anyway.
        capN = max(N, 1)
        ints = _IntPool(dev, bool(opts.get("zero_init")), _lib.COUNTS_PER_PASS * (P + 1), flags,
                        capN, P)
        counts = ints.counts

        def alloc(flag, cap):
            if not (flags & flag):
                return None, None, None
            return (torch.empty((6, cap), dtype=src.dtype, device=dev), ints.take(cap),
                    ints.take(cap))

        fin = alloc(_lib.COMPILE_FINISHED, capN)
        act = alloc(_lib.COMPILE_ACTIVE, capN * max(P, 1))
        stp = alloc(_lib.COMPILE_STOPPED, capN)
        dead = alloc(_lib.COMPILE_DEAD, capN)
        unf = torch.empty((6, capN), dtype=src.dtype, device=dev)
        unf_id = ints.take(capN)
        sc = scene.struct(face_verts)
        outs = [_ray_out(*o) for o in (fin, act, stp, dead)]
        perm = opts.get("perm")
        # an in-place trace over permuted rays compacts its ray sets in the CALLER's order itself
        # (tfrt_scene3d.ray_slot = the inverse of `perm`): nothing to restore afterwards
        own_order = bool(perm is not None and N
                         and L.tfrt_trace3d_in_place(ctypes.byref(sc), N, P) == 1)
        inv = None
        if own_order:
            inv = opts.get("ray_slot")
            inv = inverse_order(perm) if inv is None else inv
            sc.ray_slot = inv.data_ptr()
        # lazy (trace3d(lazy=True)) over an in-place trace: the forward call is given no room for
        # ray sets -- set-up launch + ONE trace launch --, and the sets are compacted from the tape
        # (tfrt_trace3d_compact into the very tensors returned here) when the caller cuts them
        in_place_now = bool(N and (own_order or (
            perm is None and L.tfrt_trace3d_in_place(ctypes.byref(sc), N, P) == 1)))
        defer = bool(opts.get("lazy") and in_place_now)
        none = [_ray_out(None, None, None) for _ in range(4)] if defer else None
        fo = none if defer else outs
        try:
            check(L.tfrt_trace3d_forward(
                _p(src), src.shape[1], N, ctypes.byref(sc), float(opts["new_ray_length"]),
                float(opts["dead_ray_length"] or 0.0), P, dt, flags,
                ctypes.byref(fo[0]), ctypes.byref(fo[1]), ctypes.byref(fo[2]),
                ctypes.byref(fo[3]), None if defer else _p(unf), None if defer else _p(unf_id),
                _p(counts), _p(ws), wsb, _stream(src)),
                "tfrt_trace3d_forward")
        finally:
            sc.ray_slot = None          # (the struct is cached on the scene)
        if defer:
            M = face_verts.shape[0]
            dead_len = float(opts["dead_ray_length"] or 0.0)

            def compact(src=src, outs=outs, unf=unf, unf_id=unf_id, counts=counts, ws=ws, inv=inv):
                check(L.tfrt_trace3d_compact(
                    _p(src), src.shape[1], N, dead_len, P, dt, flags, ctypes.byref(outs[0]),
                    ctypes.byref(outs[1]), ctypes.byref(outs[2]), ctypes.byref(outs[3]), _p(unf),
                    _p(unf_id), _p(counts), M, _p(inv), _p(ws), wsb, _stream(src)),
                    "tfrt_trace3d_compact")
            opts["_compact"] = compact
        tape = TraceTape()
        tape.src, tape.face_verts, tape.scene, tape.opts = src, face_verts, scene, dict(opts)
        tape.ws, tape.wsb, tape.counts, tape.dt = ws, wsb, counts, dt
        tape.caps = [o[0].shape[1] if o[0] is not None else 0 for o in (fin, act, stp, dead)]
        tape.dest = None
        if perm is not None and N and not own_order:
            # `src` is a permuted source (src = natural[:, perm], e.g. perm = ray_order(natural)):
            # hand every class back in the reference's order, ids in the natural numbering; the
            # reverse sweep takes the gradients back through `dest`
            tape.dest = []
            restored = []
            zero = bool(opts.get("zero_init"))
            for o, col in zip((fin, act, stp, dead),
                              (_lib.CLS_FINISHED, _lib.CLS_ACTIVE, _lib.CLS_STOPPED, _lib.CLS_DEAD)):
                if o[0] is None:
                    tape.dest.append(None)
                    restored.append(o)
                    continue
                at = _lib.COUNTS_PER_PASS * P + col
                total = counts[at:at + 1]
                inv, dest, ids_o = restore_plan(o[1], counts, P, col, perm, N, o[0].shape[1],
                                                zero=zero)
                face = torch.zeros_like(o[2]) if zero else None
                restored.append((gather_rows(o[0], inv, total), ids_o,
                                 gather_rows(o[2], inv, total, out=face)))
                tape.dest.append((dest, total))
            fin, act, stp, dead = restored
            if P > 0:
                at = _lib.COUNTS_PER_PASS * (P - 1) + _lib.CLS_ACTIVE
                total = counts[at:at + 1]
                inv, _, ids_o = restore_plan(unf_id, None, 0, None, perm, N, capN, total, zero=zero)
                unf, unf_id = gather_rows(unf, inv, total), ids_o
        ctx.tape = tape
        aux = {
            "counts": counts, "unfinished": unf, "unfinished_id": unf_id,
            "finished_id": fin[1], "finished_face": fin[2],
            "active_id": act[1], "active_face": act[2],
            "stopped_id": stp[1], "stopped_face": stp[2],
            "dead_id": dead[1], "dead_face": dead[2],
        }
        opts["_aux"] = aux
        empty = torch.empty((6, 0), dtype=src.dtype, device=dev)
        rays = [o[0] if o[0] is not None else empty for o in (fin, act, stp, dead)]
        ctx.present = [o[0] is not None for o in (fin, act, stp, dead)]
        ctx.set_materialize_grads(False)
        return _with_rows(rays, ctx.present)

    @staticmethod
    def backward(ctx, *grads):
        t = ctx.tape
        dev = t.src.device
        M = t.face_verts.shape[0]
        g_fv = torch.zeros((M, 9), dtype=torch.float64, device=dev)
        need_src = ctx.needs_input_grad[0]
        g_src = torch.zeros((6, t.src.shape[1]), dtype=torch.float64, device=dev) if need_src else None
        if t.dest is None:
            gs = _class_grads(ctx.present, t.caps, 6, dev, grads)
        else:
            # restored classes: the gradients go back into the trace's row order, row by row --
            # an error function usually touches one or two of a class's six rows (y_end, z_end),
            # and a random gather of a million float64 costs 15 us per row
            gs, at = [], 4
            for k, present in enumerate(ctx.present):
                if not present:
                    gs.append(None)
                    continue
                gb, gr = grads[k], grads[at:at + 6]
                at += 6
                d = t.dest[k]
                if gb is None and all(g is None for g in gr):
                    gs.append(None)
                    continue
                blk = torch.zeros((6, t.caps[k]), dtype=torch.float64, device=dev)
                if gb is not None:
                    gather_rows(_c(gb, torch.float64), d[0], d[1], out=blk)
                for i, g in enumerate(gr):
                    if g is None:
                        continue
                    row = gather_rows(_c(g, torch.float64), d[0], d[1],
                                      out=blk[i] if gb is None else None)
                    if gb is not None:
                        blk[i].add_(row)
                gs.append(blk)
        sc = t.scene.struct(t.face_verts)
        want_n = (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and M > 0
        g_n = torch.zeros((2, M), dtype=torch.float64, device=dev) if want_n else None
        if want_n:
            sc.grad_n_in, sc.grad_n_out = g_n[0].data_ptr(), g_n[1].data_ptr()
        try:
            check(_lib.lib().tfrt_trace3d_backward(
                _p(t.src), t.src.shape[1], t.src.shape[1], ctypes.byref(sc),
                float(t.opts["new_ray_length"]), float(t.opts["dead_ray_length"] or 0.0),
                int(t.opts["max_passes"]), t.dt,
                _p(gs[0]), t.caps[0], _p(gs[1]), t.caps[1], _p(gs[2]), t.caps[2], _p(gs[3]),
                t.caps[3], _p(g_fv), _p(g_src), _p(t.counts), _p(t.ws), t.wsb, _stream(t.src)),
                "tfrt_trace3d_backward")
        finally:
            sc.grad_n_in = sc.grad_n_out = None       # (the struct is cached on the scene)
        if g_src is not None:
            g_src = g_src.to(t.src.dtype)
        g_in = g_n[0] if (want_n and ctx.needs_input_grad[2]) else None
        g_out = g_n[1] if (want_n and ctx.needs_input_grad[3]) else None
        return g_src, g_fv, g_in, g_out, None, None


_CLASS_NAMES = ("finished", "active", "stopped", "dead")
_CLASS_NAMES_BY_COUNT_COLUMN = ("active", "finished", "stopped", "dead")   # counts[:, CLS_*]


class _LazyRows:
    """``rows[i]`` = the first n entries of row i of a class block, cut on first use."""

    def __init__(self, rows, n):
        self._rows, self._n, self._cut = rows, n, {}

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i):
        r = self._cut.get(i)
        if r is None:
            r = self._cut[i] = self._rows[i][:self._n]
        return r


def _slice_outputs(full, aux, counts, P, ncols_prefix=None):
    """Cut the full-capacity class outputs down to the per-class totals in ``counts``."""
    tail = counts[P * 8:]
    if tail[6] != 0:
        raise TfrtError("trace forward: output capacity exceeded (internal error)")
    out = {
        "counts": counts[:P * 8].reshape(P, 8).copy(),
        "counts_dev": aux["counts"],
        "n_tests": int(np.uint32(tail[4])) | (int(np.uint32(tail[5])) << 32),
        # visiting-order traces: wavefronts that were no narrow bundles (done by the grouped kernel)
        "left_over": int(tail[7]),
    }
    totals = {"active": int(tail[0]), "finished": int(tail[1]), "stopped": int(tail[2]),
              "dead": int(tail[3])}
    for name, rays in full.items():
        if aux[name + "_id"] is None:
            continue
        n = min(totals[name], rays.shape[1])
        out[name] = rays[:, :n]
        if name + "_rows" in aux:   # the same rows as separate autograd outputs (see _with_rows)
            out[name + "_rows"] = _LazyRows(aux[name + "_rows"], n)
        out[name + "_id"] = aux[name + "_id"][:n]
        out[name + "_face"] = aux[name + "_face"][:n]
    n_unf = int(out["counts"][P - 1, 0]) if P > 0 else 0
    out["unfinished"] = aux["unfinished"][:, :n_unf]
    out["unfinished_id"] = aux["unfinished_id"][:n_unf]
    return out


class PendingCounts:
    """Counts of a trace whose device->host copy is still in flight (speculative slicing)."""

    def __init__(self, host, event, full, aux, P, predicted):
        self.host, self.event, self.full, self.aux, self.P = host, event, full, aux, P
        self.predicted = predicted

    def resolve(self):
        """Wait for the copy.  Returns (prediction_was_right, outputs sliced by the true counts)."""
        self.event.synchronize()
        actual = self.host.numpy().copy()
        ok = self.predicted is not None and np.array_equal(actual, self.predicted)
        return ok, actual, _slice_outputs(self.full, self.aux, actual, self.P)


def _finish_trace(full, aux, P, predicted_counts):
    """Bring the per-class counts to the host.  Without a prediction this is the trace's one
    host sync; with a prediction (the previous step's counts) the copy is left in flight, the
    outputs are cut with the predicted sizes and ``out["pending"].resolve()`` verifies later."""
    dev_counts = aux["counts"]
    host = torch.empty(dev_counts.shape, dtype=dev_counts.dtype, pin_memory=True)
    host.copy_(dev_counts, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev_counts.device))
    pending = PendingCounts(host, ev, full, aux, P, predicted_counts)
    if predicted_counts is None:
        _, actual, out = pending.resolve()
        out["raw_counts"] = actual
        return out
    out = _slice_outputs(full, aux, predicted_counts, P)
    out["raw_counts"] = predicted_counts
    out["pending"] = pending
    return out


def trace3d(src, face_verts, scene, max_passes, new_ray_length=1.0, dead_ray_length=None,
            flags=_lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED, predicted_counts=None, perm=None,
            ray_slot=None, lazy=False):
    """Run the whole 3-D trace.  ``src`` is a (6,N) ray block (f32 or f64) on the GPU.

    Returns a dict: for each class c in finished/active/stopped/dead (when compiled) the ray
    block ``c`` (6, n_c) (differentiable w.r.t. ``face_verts`` and ``src``), ``c_id`` (source
    ray index per row) and ``c_face`` (merged face hit); ``unfinished``/``unfinished_id``;
    ``counts`` (host numpy, per pass) and ``n_tests``.  One host sync (to read the counts),
    unless ``predicted_counts`` (the ``raw_counts`` of an earlier, identical-shape trace) is
    given: then nothing blocks and ``out["pending"].resolve()`` checks the prediction.

    ``perm`` (int32, N): ``src`` is ``natural[:, perm]`` -- a source handed over in a coherent
    order (``ray_order`` / ``permute_rays``; set ``scene.coherent_rays``).  Every output then comes
    back as the trace of ``natural`` itself gives it: ids in the natural numbering, every class in
    the reference's order (restored on the device inside the autograd node, gradients included;
    an in-place trace -- ``scene.in_place`` -- compacts them in that order directly, through
    ``ray_slot`` = ``inverse_order(perm)``, computed here when not handed in).

    ``lazy``: do not wait for the counts; returns ``{"finish": callable}`` instead.
    """
    opts = dict(max_passes=max_passes, new_ray_length=new_ray_length,
                dead_ray_length=dead_ray_length, flags=flags,
                zero_init=predicted_counts is not None, perm=perm, ray_slot=ray_slot, lazy=lazy)
    grad_n = lambda t: t if (isinstance(t, torch.Tensor) and t.requires_grad) else None
    outs = _Trace3D.apply(src, face_verts, grad_n(scene.n_in_arg), grad_n(scene.n_out_arg),
                          scene, opts)
    aux = opts.pop("_aux")
    compact = opts.pop("_compact", None)
    blocks, rows = _split_rows(outs, [aux[name + "_id"] is not None for name in _CLASS_NAMES], 6)
    full = dict(zip(_CLASS_NAMES, blocks))
    for name, r in rows.items():
        aux[name + "_rows"] = r
    if lazy:
        # nothing is read back here: the caller cuts the sets (one host read of the counts) when
        # somebody asks for them -- ``out["finish"]()`` returns the dict this function returns
        P = int(max_passes)

        def finish():
            if compact is not None:     # (an in-place trace that has not compacted its sets yet)
                compact()
            return _finish_trace(full, aux, P, None)
        return {"finish": finish}
    return _finish_trace(full, aux, int(max_passes), predicted_counts)


# -------------------------------------------------------------------------------- seams

def intersect3d(rays, face_verts, intersect_epsilion=1e-10, size_epsilion=1e-10,
                ray_start_epsilion=1e-10):
    """OpticalSystem3D._intersection (engine.py:1103-1166).  rays: (6,N) block."""
    _need_gpu(rays, face_verts)
    rays = rays.contiguous()
    face_verts = _c(face_verts, torch.float64)
    N, M = rays.shape[1], face_verts.shape[0]
    dev = rays.device
    L = _lib.lib()
    wsb = L.tfrt_intersect3d_workspace_bytes(N, M)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    f = lambda: torch.empty(N, dtype=torch.float64, device=dev)
    x, y, z, ray_u, trig_u, trig_v = f(), f(), f(), f(), f(), f()
    valid = torch.empty(N, dtype=torch.uint8, device=dev)
    gather = torch.empty(N, dtype=torch.int32, device=dev)
    check(L.tfrt_intersect3d(
        _p(rays), rays.shape[1], N, _DT[rays.dtype], _p(face_verts) if M else None, M,
        float(intersect_epsilion), float(size_epsilion), float(ray_start_epsilion),
        _p(x), _p(y), _p(z), _p(valid), _p(ray_u), _p(trig_u), _p(trig_v), _p(gather),
        _p(ws), wsb, _stream(rays)), "tfrt_intersect3d")
    return x, y, z, valid.bool(), ray_u, trig_u, trig_v, gather


def ray_order(rays, face_verts=None, axis=None, return_keys=False, out=None):
    """A coherent visiting order of the rays of a (6, N) block: int32 permutation in which rays
    whose lines run close together are neighbours, so that 64 consecutive entries form a narrow
    bundle (tfrt_ray_order: Hilbert keys + a stable radix sort on the device, no host sync,
    capturable).  Trace ``permute_rays(rays, order)`` (and per-ray tables in the same order) with
    ``Scene3DArgs.coherent_rays`` / ``tfrt_scene3d.coherent_rays`` set: coherent wavefronts share
    one walk of the face hierarchy (k_intersect_beam); ``restore_order(out, order)`` gives the ray
    sets of ``rays`` itself.

    Rays that mostly share a direction are ordered along a Hilbert curve through the points where
    their lines pass the middle of the scene (``face_verts`` (M, 9): the mean centroid of 64
    sampled faces; None: the mean end point of 256 sampled rays), in the plane perpendicular to
    ``axis`` (3 numbers; None: the mean direction of the sampled rays); rays without a common
    direction (an isotropic point source) along a Hilbert curve over the octahedral map of their
    directions.  ``return_keys``: also the uint32 keys (natural order, as int32 bits); the order
    equals ``argsort(keys, stable=True)``.  ``out``: an int32 tensor of N entries to write into."""
    _need_gpu(rays, face_verts)
    if rays.dtype not in _DT:
        raise TfrtError(f"ray state dtype must be float32, float64 or float16, got {rays.dtype}")
    rays = rays.detach()
    if rays.stride(1) != 1:
        rays = rays.contiguous()
    N = rays.shape[1]
    dev = rays.device
    L = _lib.lib()
    perm = out if out is not None else torch.empty(N, dtype=torch.int32, device=dev)
    if perm.dtype != torch.int32 or perm.numel() != N or not perm.is_contiguous():
        raise TfrtError("ray_order: `out` must be a contiguous int32 tensor of N entries")
    keys = torch.empty(N, dtype=torch.int32, device=dev) if return_keys else None
    if N:
        fv = None if face_verts is None else _c(face_verts.detach(), torch.float64)
        wsb = L.tfrt_ray_order_workspace_bytes(N)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        ax = None if axis is None else (ctypes.c_double * 3)(*[float(v) for v in axis])
        check(L.tfrt_ray_order(_p(rays), rays.stride(0), N, _DT[rays.dtype], _p(fv),
                               0 if fv is None else fv.shape[0], ax, _p(perm), _p(keys), _p(ws),
                               wsb, _stream(rays)), "tfrt_ray_order")
    return (perm, keys) if return_keys else perm


def inverse_order(perm):
    """``inv`` with ``inv[perm[j]] = j`` (int32): where the caller's ray r sits in a block that was
    permuted with ``perm`` -- tfrt_scene3d.ray_slot.  (Not cached here: an order tensor is
    re-written in place by tfrt_ray_order / tfrt_source3d_order without a version bump; the
    engine keeps the inverse next to the order it belongs to.)"""
    inv = torch.empty_like(perm)
    inv[perm.long()] = torch.arange(perm.numel(), dtype=perm.dtype, device=perm.device)
    return inv


def permute_rays(rays, index, out=None):
    """``rays[:, index]`` of a (6, N) ray block (tfrt_permute_rays)."""
    _need_gpu(rays, index)
    rays = rays.detach()
    if rays.stride(1) != 1:
        rays = rays.contiguous()
    N = rays.shape[1]
    if index.dtype != torch.int32 or index.numel() != N:
        raise TfrtError("permute_rays: index must be an int32 permutation of the N rays")
    if out is None:
        out = torch.empty((6, N), dtype=rays.dtype, device=rays.device)
    if N:
        L = _lib.lib()
        wsb = L.tfrt_permute_rays_workspace_bytes(N, _DT[rays.dtype])
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=rays.device)
        check(L.tfrt_permute_rays(_p(rays), rays.stride(0), N, _DT[rays.dtype], _p(index), _p(out),
                                  out.stride(0), _p(ws), wsb, _stream(rays)), "tfrt_permute_rays")
    return out


def gather_rows(src, index, n_valid=None, out=None):
    """``src[..., index]`` for a (k, n) or (n,) tensor of 1/2/4/8-byte elements through an int32
    index (tfrt_gather_rows); ``n_valid``: int32 device scalar, entries beyond it are left alone."""
    _need_gpu(src, index)
    src = src.detach()
    one = src.dim() == 1
    s2 = src.reshape(1, -1) if one else src
    if s2.stride(1) != 1:
        s2 = s2.contiguous()
    k, n = s2.shape[0], index.numel()
    if out is None:
        out = torch.empty((k, n), dtype=src.dtype, device=src.device)
    o2 = out.reshape(1, -1) if out.dim() == 1 else out
    if n and k:
        check(_lib.lib().tfrt_gather_rows(_p(s2), s2.stride(0), k, s2.element_size(), _p(index), n,
                                          _p(n_valid), _p(o2), o2.stride(0), _stream(src)),
              "tfrt_gather_rows")
    return out.reshape(-1) if (one and out.dim() == 2) else out


def restore_plan(ids, counts_dev, P, cls_col, perm, n_src, n_rows=None, total=None, zero=False):
    """(inv, dest_of, original ids) of one output class of a trace over permuted rays
    (tfrt_restore_order): row j of the class in the reference's order = row ``inv[j]`` of the
    trace's output, ``dest_of`` is the inverse.  ``cls_col``: the class's TFRT_CLS_* column of
    ``counts_dev`` (the device counts of the trace), or None: ``ids`` is one segment (the
    unfinished set; ``total``: its row count as an int32 device scalar, default all of ``ids``).
    Entries beyond the class's row count are not written."""
    cap = ids.numel() if n_rows is None else int(n_rows)
    dev = ids.device
    new = torch.zeros if zero else torch.empty    # (zero: rows past the count must be valid indices)
    inv = new(cap, dtype=torch.int32, device=dev)
    dest = new(cap, dtype=torch.int32, device=dev)
    ids_o = new(cap, dtype=torch.int32, device=dev)
    if cap == 0:
        return inv, dest, ids_o
    L = _lib.lib()
    nseg = int(P) if cls_col is not None else 1
    wsb = L.tfrt_restore_order_workspace_bytes(int(n_src), nseg)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    if cls_col is not None:
        base = counts_dev.data_ptr()
        seg_n = ctypes.c_void_p(base + 4 * cls_col)
        seg_b = ctypes.c_void_p(base + 4 * (4 + cls_col))
        total = ctypes.c_void_p(base + 4 * (_lib.COUNTS_PER_PASS * int(P) + cls_col))
        check(L.tfrt_restore_order(_p(ids), cap, seg_n, seg_b, _lib.COUNTS_PER_PASS, nseg, total,
                                   _p(perm), int(n_src), _p(inv), _p(dest), _p(ids_o), _p(ws), wsb,
                                   _stream(ids)), "tfrt_restore_order")
    else:
        check(L.tfrt_restore_order(_p(ids), cap, None, None, 1, 1, _p(total), _p(perm), int(n_src),
                                   _p(inv), _p(dest), _p(ids_o), _p(ws), wsb, _stream(ids)),
              "tfrt_restore_order")
    return inv, dest, ids_o


def restore_order(out, perm):
    """Outputs of a trace over PERMUTED rays (``permute_rays(src, perm)``, e.g. ``perm =
    ray_order(src)``, the form tfrt_scene3d.coherent_rays wants) brought back to what the trace of
    ``src`` itself gives: ray ids mapped through ``perm``, and inside every class the rays of one
    pass sorted by ray id again (the reference's per-pass boolean_mask order,
    engine.py:2069-2111).  ``out``: the dict of trace3d(); returns a new dict.  Device kernels
    (tfrt_restore_order, tfrt_gather_rows): a counting sort by (pass, original id), no host sync."""
    n_src = perm.numel()
    counts_dev = out["counts_dev"]
    P = out["counts"].shape[0]
    new = dict(out)

    def rows(t, inv):
        if t.requires_grad:
            return t.index_select(t.dim() - 1, inv.long())
        return gather_rows(t, inv)

    for col, cls in enumerate(_CLASS_NAMES_BY_COUNT_COLUMN):
        if cls not in out or out[cls].shape[1] == 0 or out.get(cls + "_id") is None:
            continue
        n = out[cls].shape[1]
        inv, _, ids_o = restore_plan(out[cls + "_id"], counts_dev, P, col, perm, n_src, n)
        new[cls] = rows(out[cls], inv)
        new[cls + "_id"] = ids_o
        if out.get(cls + "_face") is not None:
            new[cls + "_face"] = gather_rows(out[cls + "_face"], inv)
        new.pop(cls + "_rows", None)
    if "unfinished" in out and out["unfinished"].shape[1]:
        n_u = out["unfinished"].shape[1]
        inv, _, ids_o = restore_plan(out["unfinished_id"], None, 0, None, perm, n_src, n_u)
        new["unfinished"] = rows(out["unfinished"], inv)
        new["unfinished_id"] = ids_o
    return new


def source3d_order(program, n, first=0, face_verts=None, axis=None, out=None, device=None,
                   stable=True):
    """``ray_order`` of rays ``first .. first + n`` of a source program (tfrt_source3d_order): the
    rays are never written in source order.  ``stable=False``: the same order up to ties, most
    significant digit first (tfrt_source3d_order_cells) -- fewer launches; rays with the same key may
    land in an order that varies from run to run."""
    n = int(n)
    dev = device if device is not None else (out.device if out is not None else face_verts.device)
    perm = out if out is not None else torch.empty(n, dtype=torch.int32, device=dev)
    if perm.dtype != torch.int32 or perm.numel() != n or not perm.is_contiguous():
        raise TfrtError("source3d_order: `out` must be a contiguous int32 tensor of n entries")
    _need_gpu(perm, face_verts)
    if n:
        L = _lib.lib()
        fv = None if face_verts is None else _c(face_verts.detach(), torch.float64)
        wsb = L.tfrt_ray_order_workspace_bytes(n)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        ax = None if axis is None else (ctypes.c_double * 3)(*[float(v) for v in axis])
        fn = L.tfrt_source3d_order if stable else L.tfrt_source3d_order_cells
        check(fn(ctypes.byref(program), int(first), n, _p(fv), 0 if fv is None else fv.shape[0], ax,
                 _p(perm), None, _p(ws), wsb, _stream(perm)), "tfrt_source3d_order")
    return perm


def epoch_advance(counters):
    """``c[0] += 1`` for up to 8 distinct int64 device counters in one launch (tfrt_epoch_advance):
    the distributions of a source step to their next draw."""
    counters = [c for c in counters if c is not None]
    if not counters:
        return
    _need_gpu(*counters)
    for at in range(0, len(counters), 8):
        part = counters[at:at + 8]
        arr = (ctypes.c_void_p * len(part))(*[c.data_ptr() for c in part])
        check(_lib.lib().tfrt_epoch_advance(arr, len(part), _stream(part[0])), "tfrt_epoch_advance")


def points_generate(program, n, first=0, index=None, columns=3, want_points=True, want_aux=False,
                    device=None):
    """Samples of one distribution (a ``_lib.PointsProgram``) at its current epoch
    (tfrt_points_generate): (points (n, columns) f64 or None, aux0, aux1 (n,) f64 or None)."""
    dev = device if device is not None else (index.device if index is not None else None)
    n = int(n)
    pts = torch.empty((n, columns), dtype=torch.float64, device=dev) if want_points else None
    a0 = torch.empty(n, dtype=torch.float64, device=dev) if want_aux else None
    a1 = torch.empty(n, dtype=torch.float64, device=dev) if want_aux else None
    ref = pts if pts is not None else a0
    if n and ref is not None:
        _need_gpu(ref)
        check(_lib.lib().tfrt_points_generate(ctypes.byref(program), _p(index), int(first), n,
                                              _p(pts), int(columns), _p(a0), _p(a1), _stream(ref)),
              "tfrt_points_generate")
    return pts, a0, a1


def source3d_generate(program, n, dtype=None, first=0, index=None, rays_out=None, fields=False,
                      device=None):
    """Rays of a source program (``_lib.Source3DProgram``) at the current epochs of its
    distributions (tfrt_source3d_generate): (ray block (6, n) of ``dtype`` or None, fields (6, n)
    f64 or None); ``index``: the rays ``first + index[j]`` instead of ``first + j``."""
    n = int(n)
    dev = device if device is not None else (index.device if index is not None else
                                             (rays_out.device if rays_out is not None else None))
    rays = rays_out
    if rays is None and dtype is not None:
        rays = torch.empty((6, n), dtype=dtype, device=dev)
    fl = torch.empty((6, n), dtype=torch.float64, device=dev) if fields else None
    ref = rays if rays is not None else fl
    if n and ref is not None:
        _need_gpu(ref)
        dt = _DT[rays.dtype] if rays is not None else _lib.F64
        check(_lib.lib().tfrt_source3d_generate(
            ctypes.byref(program), _p(index), int(first), n, dt, _p(rays),
            rays.stride(0) if rays is not None else 0, _p(fl), n if fl is not None else 0,
            _stream(ref)), "tfrt_source3d_generate")
    return rays, fl


def morton_order(face_verts):
    """Permutation of the faces by the Morton code of their centroids (30-bit, 10 bits per
    axis): spatially close faces become neighbours.  Pass it as ``cluster_order``."""
    fv = face_verts.detach()
    c = (fv[:, 0:3] + fv[:, 3:6] + fv[:, 6:9]) / 3.0
    lo, hi = c.min(dim=0).values, c.max(dim=0).values
    q = ((c - lo) / torch.clamp(hi - lo, min=1e-300) * 1023.0).clamp(0, 1023).to(torch.int64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(key, stable=True).to(torch.int32)


def cluster_order(face_verts, leaf=16, group=128):
    """Permutation of the faces that makes every aligned run of ``leaf`` consecutive entries
    (and of ``group`` entries: the kernels' clusters and superclusters) a compact patch:
    recursive median split of the face centroids along the longest axis of their bounding box,
    the left part a multiple of ``group`` (of ``leaf`` below that), down to parts of <= ``leaf``
    faces; outsized faces (a target plane behind a fine lens mesh, ...) go last.  Pass it as
    ``cluster_order``.  Unlike a Morton order there are no space-filling-curve jumps inside a run,
    so the clusters' bounding spheres stay small (measured on the cfg4 lens: 10 cluster hits per
    ray instead of 17).  On the device (tfrt_cluster_order: a radix sort per tree level, no host
    sync), once per mesh topology."""
    _need_gpu(face_verts)
    fv = _c(face_verts.detach(), torch.float64)
    M = fv.shape[0]
    order = torch.empty(M, dtype=torch.int32, device=fv.device)
    if M:
        L = _lib.lib()
        wsb = L.tfrt_cluster_order_workspace_bytes(M, int(leaf), int(group))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=fv.device)
        check(L.tfrt_cluster_order(_p(fv), M, int(leaf), int(group), _p(order), _p(ws), wsb,
                                   _stream(fv)), "tfrt_cluster_order")
    return order


def snell3d(x_start, y_start, z_start, x_end, y_end, z_end, norm, n_in, n_out, new_ray_length):
    """geometry.snells_law_3D (geometry.py:671-753) -> (6,n) f64 block of the new rays."""
    args = [_c(a, torch.float64) for a in (x_start, y_start, z_start, x_end, y_end, z_end, norm)]
    _need_gpu(*args)
    n = args[0].shape[0]
    n_in = _c(torch.as_tensor(n_in, dtype=torch.float64, device=args[0].device).expand(n), torch.float64)
    n_out = _c(torch.as_tensor(n_out, dtype=torch.float64, device=args[0].device).expand(n), torch.float64)
    out = torch.empty((6, n), dtype=torch.float64, device=args[0].device)
    check(_lib.lib().tfrt_snell3d(n, *[_p(a) for a in args], _p(n_in), _p(n_out),
                                  float(new_ray_length), _p(out), _stream(out)), "tfrt_snell3d")
    return out


def snell2d(x_start, y_start, x_end, y_end, norm, n_in, n_out, new_ray_length):
    """geometry.snells_law_2D (geometry.py:565-653) -> (4,n) f64 block of the new rays."""
    args = [_c(a, torch.float64) for a in (x_start, y_start, x_end, y_end, norm)]
    _need_gpu(*args)
    n = args[0].shape[0]
    n_in = _c(torch.as_tensor(n_in, dtype=torch.float64, device=args[0].device).expand(n), torch.float64)
    n_out = _c(torch.as_tensor(n_out, dtype=torch.float64, device=args[0].device).expand(n), torch.float64)
    out = torch.empty((4, n), dtype=torch.float64, device=args[0].device)
    check(_lib.lib().tfrt_snell2d(n, *[_p(a) for a in args], _p(n_in), _p(n_out),
                                  float(new_ray_length), _p(out), _stream(out)), "tfrt_snell2d")
    return out


# ================================================================================= 2-D

from ._lib import Scene2D  # noqa: E402


class Scene2DArgs:
    """Device tensors describing the merged 2-D boundary sets (tfrt_scene2d).  ``segments`` /
    ``arcs`` are the dicts built by ``OpticalSystem2D._merge_kind`` (``geo`` (M,4|5) f64,
    ``cat`` int32, optional ``mat_in/mat_out/n_in/n_out``) or None."""

    def __init__(self, segments, arcs, n_table, index_mode, ghost, intersect_epsilion=1e-10,
                 size_epsilion=1e-10, ray_start_epsilion=1e-10, finite_tir_gradient=False):
        self.segments, self.arcs = segments, arcs
        self.finite_tir_gradient = bool(finite_tir_gradient)
        self.n_table = _c(n_table, torch.float64)
        self.index_mode, self.ghost = index_mode, ghost
        self.eps = (float(intersect_epsilion), float(size_epsilion), float(ray_start_epsilion))
        self._keep = []

    def struct(self, seg_geo, arc_geo):
        sc = Scene2D()
        self._keep = []

        def fill(prefix, info, geo, count_field):
            n = 0 if geo is None else geo.shape[0]
            setattr(sc, count_field, n)
            names = {"seg": ("seg", "seg_cat", "seg_mat_in", "seg_mat_out", "seg_n_in", "seg_n_out"),
                     "arc": ("arc", "arc_cat", "arc_mat_in", "arc_mat_out", "arc_n_in", "arc_n_out")}[prefix]
            for nm in names:
                setattr(sc, nm, None)
            if n == 0:
                return
            setattr(sc, names[0], geo.data_ptr())
            setattr(sc, names[1], info["cat"].data_ptr())
            if self.ghost:
                ones = torch.ones(n, dtype=torch.float64, device=geo.device)
                self._keep.append(ones)
                setattr(sc, names[4], ones.data_ptr())
                setattr(sc, names[5], ones.data_ptr())
            elif self.index_mode:
                if info["mat_in"] is None or info["mat_out"] is None:
                    raise TfrtError("StandardReaction('index') needs mat_in / mat_out on every "
                                    "optical boundary")
                setattr(sc, names[2], info["mat_in"].data_ptr())
                setattr(sc, names[3], info["mat_out"].data_ptr())
            else:
                if info["n_in"] is None or info["n_out"] is None:
                    raise TfrtError("StandardReaction('value') needs n_in / n_out on every optical "
                                    "boundary")
                setattr(sc, names[4], info["n_in"].data_ptr())
                setattr(sc, names[5], info["n_out"].data_ptr())

        fill("seg", self.segments, seg_geo, "n_segments")
        fill("arc", self.arcs, arc_geo, "n_arcs")
        if self.n_table is not None and self.n_table.numel() and self.index_mode and not self.ghost:
            sc.n_table = self.n_table.data_ptr()
            sc.n_table_stride = self.n_table.shape[1]
            sc.n_materials = self.n_table.shape[0]
        else:
            sc.n_table, sc.n_table_stride, sc.n_materials = None, 0, 0
        sc.intersect_epsilion, sc.size_epsilion, sc.ray_start_epsilion = self.eps
        sc.finite_tir_gradient = 1 if self.finite_tir_gradient else 0
        return sc


class _Trace2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, seg_geo, arc_geo, scene, opts):
        _need_gpu(src, seg_geo, arc_geo)
        dev = src.device
        if src.dtype not in _DT:
            raise TfrtError(f"ray state dtype must be float32, float64 or float16, got {src.dtype}")
        src = src.contiguous()
        seg_geo = _c(seg_geo, torch.float64)
        arc_geo = _c(arc_geo, torch.float64)
        N = src.shape[1]
        P = int(opts["max_passes"])
        flags = int(opts["flags"])
        dt = _DT[src.dtype]
        L = _lib.lib()
        Ms = 0 if seg_geo is None else seg_geo.shape[0]
        Ma = 0 if arc_geo is None else arc_geo.shape[0]
        wsb = L.tfrt_trace2d_workspace_bytes(N, Ms, Ma, P, dt)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        capN = max(N, 1)
        ints = _IntPool(dev, bool(opts.get("zero_init")), _lib.COUNTS_PER_PASS * (P + 1), flags,
                        capN, P)
        counts = ints.counts

        def alloc(flag, cap):
            if not (flags & flag):
                return None, None, None
            return (torch.empty((4, cap), dtype=src.dtype, device=dev),  # (see _Trace3D.forward)
                    ints.take(cap), ints.take(cap))

        fin = alloc(_lib.COMPILE_FINISHED, capN)
        act = alloc(_lib.COMPILE_ACTIVE, capN * max(P, 1))
        stp = alloc(_lib.COMPILE_STOPPED, capN)
        dead = alloc(_lib.COMPILE_DEAD, capN)
        unf = torch.empty((4, capN), dtype=src.dtype, device=dev)
        unf_id = ints.take(capN)
        sc = scene.struct(seg_geo, arc_geo)
        outs = [_ray_out(*o) for o in (fin, act, stp, dead)]
        check(L.tfrt_trace2d_forward(
            _p(src), src.shape[1], N, ctypes.byref(sc), float(opts["new_ray_length"]),
            float(opts["dead_ray_length"] or 0.0), P, dt, flags,
            ctypes.byref(outs[0]), ctypes.byref(outs[1]), ctypes.byref(outs[2]),
            ctypes.byref(outs[3]), _p(unf), _p(unf_id), _p(counts), _p(ws), wsb, _stream(src)),
            "tfrt_trace2d_forward")
        tape = TraceTape()
        tape.src, tape.seg, tape.arc, tape.scene, tape.opts = src, seg_geo, arc_geo, scene, dict(opts)
        tape.ws, tape.wsb, tape.counts, tape.dt = ws, wsb, counts, dt
        tape.caps = [o[0].shape[1] if o[0] is not None else 0 for o in (fin, act, stp, dead)]
        ctx.tape = tape
        opts["_aux"] = {
            "counts": counts, "unfinished": unf, "unfinished_id": unf_id,
            "finished_id": fin[1], "finished_face": fin[2], "active_id": act[1],
            "active_face": act[2], "stopped_id": stp[1], "stopped_face": stp[2],
            "dead_id": dead[1], "dead_face": dead[2],
        }
        empty = torch.empty((4, 0), dtype=src.dtype, device=dev)
        ctx.present = [o[0] is not None for o in (fin, act, stp, dead)]
        ctx.set_materialize_grads(False)
        return _with_rows([o[0] if o[0] is not None else empty for o in (fin, act, stp, dead)],
                          ctx.present)

    @staticmethod
    def backward(ctx, *grads):
        t = ctx.tape
        dev = t.src.device
        g_seg = None if t.seg is None else torch.zeros_like(t.seg)
        g_arc = None if t.arc is None else torch.zeros_like(t.arc)
        need_src = ctx.needs_input_grad[0]
        g_src = torch.zeros((4, t.src.shape[1]), dtype=torch.float64, device=dev) if need_src else None
        gs = _class_grads(ctx.present, t.caps, 4, dev, grads)
        sc = t.scene.struct(t.seg, t.arc)
        check(_lib.lib().tfrt_trace2d_backward(
            _p(t.src), t.src.shape[1], t.src.shape[1], ctypes.byref(sc),
            float(t.opts["new_ray_length"]), float(t.opts["dead_ray_length"] or 0.0),
            int(t.opts["max_passes"]), t.dt,
            _p(gs[0]), t.caps[0], _p(gs[1]), t.caps[1], _p(gs[2]), t.caps[2], _p(gs[3]), t.caps[3],
            _p(g_seg), _p(g_arc), _p(g_src), _p(t.counts), _p(t.ws), t.wsb, _stream(t.src)),
            "tfrt_trace2d_backward")
        if g_src is not None:
            g_src = g_src.to(t.src.dtype)
        return g_src, g_seg, g_arc, None, None


def trace2d(src, scene, max_passes, new_ray_length=1.0, dead_ray_length=None,
            flags=_lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED, predicted_counts=None):
    """Whole 2-D trace; same result dict as ``trace3d`` with (4, n) ray blocks.  The ``*_face``
    arrays index the merged segments first and then ``n_segments + merged arc index``."""
    opts = dict(max_passes=max_passes, new_ray_length=new_ray_length,
                dead_ray_length=dead_ray_length, flags=flags,
                zero_init=predicted_counts is not None)
    seg_geo = None if scene.segments is None else scene.segments["geo"]
    arc_geo = None if scene.arcs is None else scene.arcs["geo"]
    outs = _Trace2D.apply(src, seg_geo, arc_geo, scene, opts)
    aux = opts.pop("_aux")
    blocks, rows = _split_rows(outs, [aux[name + "_id"] is not None for name in _CLASS_NAMES], 4)
    full = dict(zip(_CLASS_NAMES, blocks))
    for name, r in rows.items():
        aux[name + "_rows"] = r
    out = _finish_trace(full, aux, int(max_passes), predicted_counts)
    out["n_segments"] = 0 if seg_geo is None else seg_geo.shape[0]
    return out


def _seam2d(fn_name, rays, prim, eps):
    _need_gpu(rays, prim)
    rays = rays.contiguous()
    prim = _c(prim, torch.float64)
    N, M = rays.shape[1], prim.shape[0]
    dev = rays.device
    f = lambda: torch.empty(N, dtype=torch.float64, device=dev)
    x, y, ray_u, prim_u = f(), f(), f(), f()
    valid = torch.empty(N, dtype=torch.uint8, device=dev)
    gather = torch.empty(N, dtype=torch.int32, device=dev)
    check(getattr(_lib.lib(), fn_name)(
        _p(rays), rays.shape[1], N, _DT[rays.dtype], _p(prim) if M else None, M,
        float(eps[0]), float(eps[1]), float(eps[2]), _p(x), _p(y), _p(valid), _p(ray_u),
        _p(prim_u), _p(gather), _stream(rays)), fn_name)
    return x, y, valid.bool(), ray_u, prim_u, gather


def segment_intersection(rays, seg, intersect_epsilion=1e-10, size_epsilion=1e-10,
                         ray_start_epsilion=1e-10):
    """OpticalSystem2D._segment_intersection (engine.py:688-749); rays (4,N), seg (M,4)."""
    return _seam2d("tfrt_segment_intersection", rays, seg,
                   (intersect_epsilion, size_epsilion, ray_start_epsilion))


def arc_intersection(rays, arc, intersect_epsilion=1e-10, size_epsilion=1e-10,
                     ray_start_epsilion=1e-10):
    """OpticalSystem2D._arc_intersection (engine.py:768-866); rays (4,N), arc (M,5)."""
    return _seam2d("tfrt_arc_intersection", rays, arc,
                   (intersect_epsilion, size_epsilion, ray_start_epsilion))


def gather_optical_2d(system, out):
    """Boundary data of the reacting rays of a 2-D pass, in the order of ``out['active']``
    (segment-hit rays then arc-hit rays), correctly paired (the reference's mixed-system
    concat pairs them wrongly, engine.py:1958-1965)."""
    face = out["active_face"].long()
    ns = out["n_segments"]
    result = {}
    seg_set = system._amalgamated_optical_segments
    arc_set = system._amalgamated_optical_arcs
    is_arc = face >= ns
    geo = {"x_start", "y_start", "x_end", "y_end", "x_center", "y_center", "angle_start",
           "angle_end", "radius"}
    keys = None
    for s in (seg_set, arc_set):
        if bool(s):
            k = set(s.keys()) - geo
            keys = k if keys is None else (keys & k)
    for f in (keys or ()):
        parts = []
        if bool(seg_set):
            parts.append(seg_set[f][face[~is_arc]])
        if bool(arc_set):
            parts.append(arc_set[f][face[is_arc] - ns])
        result[f] = torch.cat(parts) if len(parts) > 1 else parts[0]
    return result
