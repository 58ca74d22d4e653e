"""
CPU float64 restatement of the tfrt trace loop (TEST INFRASTRUCTURE -- see
oracle/__init__.py).  Follows, in order:

* tfrt/boundaries.py:890-923   ``update_fields_from_vertices``  -> ``faces_from_vertices``
* tfrt/engine.py:50-76         ``amalgamate``                   -> ``amalgamate``
* tfrt/engine.py:971-1018      ``_merge_boundaries`` (3-D)      -> ``merge_boundaries``
* tfrt/engine.py:1103-1166     ``OpticalSystem3D._intersection``-> ``intersection_3d``
* tfrt/engine.py:688-749       ``_segment_intersection``        -> ``segment_intersection``
* tfrt/engine.py:768-866       ``_arc_intersection``            -> ``arc_intersection``
* tfrt/engine.py:627-657       ``_seg_or_arc``                  -> ``seg_or_arc``
* tfrt/engine.py:667-670       ``_get_arc_norm``                -> ``get_arc_norm``
* tfrt/engine.py:1988-2191     ``process_projection_3D``        -> ``_project_3d``
* tfrt/engine.py:1544-1986     ``process_projection_2D``        -> ``_project_2d``
* tfrt/operation.py:255-307    ``StandardReaction.main``        -> ``_react``
* tfrt/engine.py:2193-2330     ``single_pass`` / ``ray_trace``  -> ``single_pass`` / ``ray_trace``
* tfrt/materials.py:25-104                                      -> ``MATERIALS``

A "ray set" / "boundary set" is a dict of 1-D float64 tensors of equal length (plus
``norm`` of shape (M, 3) for triangles), exactly like the reference's field dicts.
"""
import math

import torch

from . import geom

F64 = torch.float64
I64 = torch.int64
PI = math.pi
OPTICAL, STOP, TARGET = 0, 1, 2

SEGMENT_GEO = ("x_start", "y_start", "x_end", "y_end")
ARC_GEO = ("x_center", "y_center", "angle_start", "angle_end", "radius")
TRIANGLE_GEO = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2", "norm")


# ------------------------------------------------------------------------ materials

def _const(n):
    return lambda x: n * torch.ones_like(x)


def acrylic(x):
    return geom.sqrt(
        2.1778 + 6.1209e-9 * x ** 2 - 1.5004e-15 * x ** 4 + 2.3678e4 * x ** -2
        - 4.2137e9 * x ** -4 + 7.3417e14 * x ** -6 - 4.5042e19 * x ** -8
    )


def _sellmeier(terms):
    def f(x):
        acc = 1
        for b, c in terms:
            acc = acc + b * x ** 2 / (x ** 2 - c)
        return geom.sqrt(acc)
    return f


MATERIALS = {
    "vacuum": lambda x: torch.ones_like(x),
    "reflective": lambda x: torch.zeros_like(x),
    "acrylic": acrylic,
    "crown_glass": _sellmeier(
        [(1.1273555e0, 7.20341707e3), (1.24412303e-1, 2.69835916e4), (8.27100531e-1, 1.00384588e8)]
    ),
    "flint_glass": _sellmeier(
        [(1.34533359e0, 9.97743871e3), (2.09073176e-1, 4.70450767e4), (9.37357162e-1, 1.11886764e8)]
    ),
    "fused_silica": _sellmeier(
        [(6.961663e-1, 4.679148e3), (4.079426e-1, 1.3512063e4), (8.974794e-1, 9.7934002538e7)]
    ),
    "polycarbonate": _sellmeier([(1.4182e0, 2.1304e4)]),
    "soda_lime": lambda x: 1.5130e0 - 3.169e-9 * x ** 2 + 3.962e3 * x ** -2,
}
build_constant_material = _const


# ------------------------------------------------------------------------- plumbing

def amalgamate(stuff, signature=None):
    """engine.py:50-76: concat every common field of the non-empty sets."""
    items = [s for s in stuff if s]
    if not items:
        return {}
    if not signature:
        signature = None
        for s in items:
            keys = set(s.keys())
            signature = keys if signature is None else (signature & keys)
    return {f: torch.cat([s[f] for s in items], 0) for f in signature}


def _mask(rays, m):
    """tf.boolean_mask on every field (stable)."""
    return {f: v[m] for f, v in rays.items()}


def faces_from_vertices(vertices, faces, vertex_update_map=None):
    """boundaries.py:890-923.  vertices (V,3) f64, faces (F,3) int -> triangle fields.

    ``vertex_update_map`` (F,3) bool: corners that are False get ``stop_gradient``.
    """
    faces = torch.as_tensor(faces, dtype=I64)
    pts = [vertices[faces[:, c]] for c in range(3)]
    if vertex_update_map is not None:
        m = torch.as_tensor(vertex_update_map, dtype=torch.bool)
        pts = [torch.where(m[:, c:c + 1], p, p.detach()) for c, p in enumerate(pts)]
    first, second, third = pts
    # tf.linalg.cross (boundaries.py:919): products and differences rounded one by one.  Spelled
    # out because this image's torch.linalg.cross kernel is compiled with fused multiply-adds
    # (one rounding fewer per component: half of all normals then differ in the last bit).
    a, b = second - first, third - second
    cross = torch.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                         a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                         a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], dim=1)
    # tf.linalg.normalize (boundaries.py:918): x / sqrt(reduce_sum(x * x)), correctly rounded sqrt
    norm = cross / geom.sqrt(torch.sum(cross * cross, dim=1, keepdim=True))
    out = {"norm": norm}
    for name, p in zip(("p", "1", "2"), pts):
        out["x" + name], out["y" + name], out["z" + name] = p[:, 0], p[:, 1], p[:, 2]
    return out


def _tag(bset, cat, shape_field):
    if not bset:
        return {}
    bset = dict(bset)
    bset["catagory"] = cat * torch.ones_like(bset[shape_field], dtype=I64)
    return bset


def merge_boundaries(optical, stop, target, geo=TRIANGLE_GEO):
    """engine.py:971-1018 (and the 2-D twins 418-521): order optical, stop, target."""
    shape_field = geo[0]
    sets = [_tag(optical, OPTICAL, shape_field), _tag(stop, STOP, shape_field),
            _tag(target, TARGET, shape_field)]
    merged = amalgamate(sets, set(geo) | {"catagory"})
    counts = [0 if not s else int(s[shape_field].shape[0]) for s in (optical, stop, target)]
    return merged, counts


# ------------------------------------------------------------------- nearest-hit 3D

def _nearest(valid, ray_u, *fields):
    """engine.py:1143-1164: sentinel fill, argmin over boundaries, gather."""
    inf = 2 * torch.max(ray_u) * torch.ones_like(ray_u)
    ray_u = torch.where(valid, ray_u, inf)
    closest = torch.argmin(ray_u, dim=0)
    any_valid = torch.any(valid, dim=0)
    cols = torch.arange(ray_u.shape[1], dtype=I64)
    gathered = [f[closest, cols] for f in (ray_u,) + fields]
    return any_valid, closest, cols, gathered


def intersection_3d(rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2,
                    intersect_epsilion, size_epsilion, ray_start_epsilion, chunk=2048):
    """engine.py:1103-1166.  Returns x,y,z,valid,ray_u,trig_u,trig_v,gather_ray,gather_trig.

    Chunked over rays; the only cross-ray quantity in the reference is the sentinel
    ``2*reduce_max(ray_u)``, which acts as +inf (any valid ray_u is >= ray_start_epsilion > 0
    and < 2*max), so per-chunk evaluation gives the same result.
    """
    n = int(rx1.shape[0])
    outs = [[] for _ in range(9)]
    for lo in range(0, max(n, 1), chunk):
        sl = slice(lo, min(lo + chunk, n))
        x, y, z, valid, ray_u, trig_u, trig_v = geom.line_triangle_intersect(
            rx1[sl], ry1[sl], rz1[sl], rx2[sl], ry2[sl], rz2[sl],
            xp, yp, zp, x1, y1, z1, x2, y2, z2, intersect_epsilion,
        )
        valid = valid & (trig_u >= -size_epsilion)
        valid = valid & (trig_v >= -size_epsilion)
        valid = valid & (trig_u + trig_v <= 1 + size_epsilion)
        valid = valid & (ray_u >= ray_start_epsilion)
        if ray_u.numel() == 0:
            continue
        any_valid, closest, cols, (ru, gx, gy, gz, tu, tv) = _nearest(
            valid, ray_u, x, y, z, trig_u, trig_v)
        for o, v in zip(outs, (gx, gy, gz, any_valid, ru, tu, tv, cols + lo, closest)):
            o.append(v)
    if not outs[0]:
        e = torch.zeros(0, dtype=F64)
        ei = torch.zeros(0, dtype=I64)
        return e, e, e, torch.zeros(0, dtype=torch.bool), e, e, e, ei, ei
    return tuple(torch.cat(o) for o in outs)


# ------------------------------------------------------------------- nearest-hit 2D

def segment_intersection(rx1, ry1, rx2, ry2, sx1, sy1, sx2, sy2,
                         intersect_epsilion, size_epsilion, ray_start_epsilion):
    """engine.py:688-749.  Returns x, y, valid, ray_u, seg_u, gather_ray, gather_segment."""
    x, y, valid, ray_u, seg_u = geom.line_intersect(
        rx1, ry1, rx2, ry2, sx1, sy1, sx2, sy2, intersect_epsilion)
    valid = valid & (seg_u >= -size_epsilion)
    valid = valid & (seg_u <= 1 + size_epsilion)
    valid = valid & (ray_u >= ray_start_epsilion)
    any_valid, closest, cols, (ru, gx, gy, su) = _nearest(valid, ray_u, x, y, seg_u)
    return gx, gy, any_valid, ru, su, cols, closest


def arc_intersection(rx1, ry1, rx2, ry2, xc, yc, a1, a2, r,
                     intersect_epsilion, size_epsilion, ray_start_epsilion):
    """engine.py:768-866.  Returns x, y, valid, ray_u, arc_u, gather_ray, gather_arc."""
    plus, minus = geom.line_circle_intersect(rx1, ry1, rx2, ry2, xc, yc, r, intersect_epsilion)
    plus["valid"] = plus["valid"] & (plus["u"] >= ray_start_epsilion)
    minus["valid"] = minus["valid"] & (minus["u"] >= ray_start_epsilion)
    a1 = torch.as_tensor(a1, dtype=F64).reshape(-1, 1)
    a2 = torch.as_tensor(a2, dtype=F64).reshape(-1, 1)
    plus["valid"] = plus["valid"] & geom.angle_in_interval(plus["v"], a1, a2)
    minus["valid"] = minus["valid"] & geom.angle_in_interval(minus["v"], a1, a2)

    inf = 2 * torch.max(plus["u"]) * torch.ones_like(plus["u"])
    plus["u"] = torch.where(plus["valid"], plus["u"], inf)
    minus["u"] = torch.where(minus["valid"], minus["u"], inf)

    choose_minus = minus["u"] < plus["u"]
    minus["valid"] = minus["valid"] & choose_minus
    plus["valid"] = plus["valid"] & ~choose_minus
    valid = minus["valid"] | plus["valid"]
    x = torch.where(choose_minus, minus["x"], plus["x"])
    y = torch.where(choose_minus, minus["y"], plus["y"])
    ray_u = torch.where(choose_minus, minus["u"], plus["u"])
    arc_u = torch.where(choose_minus, minus["v"], plus["v"])

    closest = torch.argmin(ray_u, dim=0)
    any_valid = torch.any(valid, dim=0)
    cols = torch.arange(ray_u.shape[1], dtype=I64)
    return (x[closest, cols], y[closest, cols], any_valid, ray_u[closest, cols],
            arc_u[closest, cols], cols, closest)


def seg_or_arc(seg_u, arc_u, seg_valid, arc_valid):
    """engine.py:652-657."""
    has_both = seg_valid & arc_valid
    seg_less = seg_u < arc_u
    return torch.where(has_both, seg_less, seg_valid), torch.where(has_both, ~seg_less, arc_valid)


def get_arc_norm(radius, arc_u, gather_arc):
    """engine.py:667-670."""
    radius = radius[gather_arc]
    arc_norm = torch.where(radius < 0, arc_u + PI, arc_u)
    return torch.remainder(arc_norm + PI, 2 * PI) - PI


# --------------------------------------------------------------------------- system

class System:
    """Plain container standing in for OpticalSystem2D/3D after ``update()``.

    3-D: ``optical`` / ``stop`` / ``target`` are triangle field dicts (or None).
    2-D: ``optical_segments``, ``optical_arcs``, ``stop_segments``, ... likewise.
    ``materials`` is a list of callables wavelength[nm] -> n (the reference wraps them as
    ``{"n": f}`` dicts; only ``"n"`` is ever read, operation.py:266).
    """

    def __init__(self, dimension, materials=(), intersect_epsilion=1e-10,
                 size_epsilion=1e-10, ray_start_epsilion=1e-10, **sets):
        self.dimension = dimension
        self.materials = list(materials)
        self.eps = (intersect_epsilion, size_epsilion, ray_start_epsilion)
        names = (("optical", "stop", "target") if dimension == 3 else
                 ("optical_segments", "stop_segments", "target_segments",
                  "optical_arcs", "stop_arcs", "target_arcs"))
        for n in names:
            setattr(self, n, sets.pop(n, None) or {})
        assert not sets, f"unknown sets {list(sets)}"
        if dimension == 3:
            self.merged, self.counts = merge_boundaries(self.optical, self.stop, self.target)
        else:
            self.merged_segments, self.seg_counts = merge_boundaries(
                self.optical_segments, self.stop_segments, self.target_segments, SEGMENT_GEO)
            self.merged_arcs, self.arc_counts = merge_boundaries(
                self.optical_arcs, self.stop_arcs, self.target_arcs, ARC_GEO)


def _gather_set(bset, idx):
    return {f: v[idx] for f, v in bset.items()}


def _project_3d(system, rays, flags, history, chunk):
    """engine.py:1988-2191.  Mutates ``rays`` end points (as the reference does)."""
    ie, se, rse = system.eps
    m = system.merged
    x, y, z, valid, ray_u, trig_u, trig_v, gather_ray, gather_trig = intersection_3d(
        rays["x_start"], rays["y_start"], rays["z_start"],
        rays["x_end"], rays["y_end"], rays["z_end"],
        m["xp"], m["yp"], m["zp"], m["x1"], m["y1"], m["z1"], m["x2"], m["y2"], m["z2"],
        ie, se, rse, chunk=chunk)
    norm = m["norm"][gather_trig]
    result = {"rays": {}}

    if flags["compile_dead_rays"]:
        dead = _mask(rays, ~valid)
        dl = flags.get("dead_ray_length")
        if dl:
            for a in "xyz":
                dead[a + "_end"] = dead[a + "_start"] + dl * (dead[a + "_end"] - dead[a + "_start"])
        history["dead"].append(dead)
        result["rays"]["dead"] = dead

    rays["x_end"] = torch.where(valid, x, rays["x_end"])
    rays["y_end"] = torch.where(valid, y, rays["y_end"])
    rays["z_end"] = torch.where(valid, z, rays["z_end"])

    btype = m["catagory"][gather_trig]
    is_active = valid & (btype == OPTICAL)
    active = _mask(rays, is_active)
    if flags["compile_active_rays"]:
        history["active"].append(active)
    result["rays"]["active"] = active

    if flags["compile_finished_rays"]:
        fin = _mask(rays, valid & (btype == TARGET))
        history["finished"].append(fin)
        result["rays"]["finished"] = fin
    if flags["compile_stopped_rays"]:
        stp = _mask(rays, valid & (btype == STOP))
        history["stopped"].append(stp)
        result["rays"]["stopped"] = stp

    gather_optical = gather_trig[is_active]
    optical = _gather_set(system.optical, gather_optical) if system.optical else {}
    optical["norm"] = norm[is_active]
    result["optical"] = optical
    result["gather_trig"] = gather_trig
    result["valid"] = valid
    return result


def _project_2d(system, rays, flags, history, bug_compatible):
    """engine.py:1544-1986.

    In a mixed segment+arc system the reference concatenates active *rays* as [seg, arc]
    but their *optical data* as [arc, seg] (engine.py:1958-1965).  ``bug_compatible=True``
    reproduces that; the default pairs them correctly (what the build does).
    """
    ie, se, rse = system.eps
    has_seg = bool(system.merged_segments)
    has_arc = bool(system.merged_arcs)
    seg, arc = {}, {}
    if has_seg:
        ms = system.merged_segments
        (seg["x"], seg["y"], seg["valid"], seg["ray_u"], seg["segment_u"], _,
         seg["gather"]) = segment_intersection(
            rays["x_start"], rays["y_start"], rays["x_end"], rays["y_end"],
            ms["x_start"], ms["y_start"], ms["x_end"], ms["y_end"], ie, se, rse)
        seg["norm"] = (torch.atan2(ms["y_end"] - ms["y_start"], ms["x_end"] - ms["x_start"])
                       + PI / 2.0)[seg["gather"]]
    if has_arc:
        ma = system.merged_arcs
        (arc["x"], arc["y"], arc["valid"], arc["ray_u"], arc["arc_u"], _,
         arc["gather"]) = arc_intersection(
            rays["x_start"], rays["y_start"], rays["x_end"], rays["y_end"],
            ma["x_center"], ma["y_center"], ma["angle_start"], ma["angle_end"], ma["radius"],
            ie, se, rse)
        arc["norm"] = get_arc_norm(ma["radius"], arc["arc_u"], arc["gather"])
    if has_seg and has_arc:
        seg["valid"], arc["valid"] = seg_or_arc(seg["ray_u"], arc["ray_u"], seg["valid"], arc["valid"])

    result = {"rays": {}}
    if flags["compile_dead_rays"]:
        hit = torch.zeros_like(rays["x_start"], dtype=torch.bool)
        if has_seg:
            hit = hit | seg["valid"]
        if has_arc:
            hit = hit | arc["valid"]
        dead = _mask(rays, ~hit)
        dl = flags.get("dead_ray_length")
        if dl:
            for a in "xy":
                dead[a + "_end"] = dead[a + "_start"] + dl * (dead[a + "_end"] - dead[a + "_start"])
        history["dead"].append(dead)
        result["rays"]["dead"] = dead

    per_kind = {}
    for kind, proj, merged, opt_set in (
        ("seg", seg, getattr(system, "merged_segments", {}), system.optical_segments),
        ("arc", arc, getattr(system, "merged_arcs", {}), system.optical_arcs),
    ):
        if not proj:
            continue
        rays["x_end"] = torch.where(proj["valid"], proj["x"], rays["x_end"])
        rays["y_end"] = torch.where(proj["valid"], proj["y"], rays["y_end"])
        btype = merged["catagory"][proj["gather"]]
        is_active = proj["valid"] & (btype == OPTICAL)
        entry = {"active": _mask(rays, is_active)}
        if flags["compile_active_rays"]:
            history["active"].append(entry["active"])
        if flags["compile_finished_rays"]:
            entry["finished"] = _mask(rays, proj["valid"] & (btype == TARGET))
            history["finished"].append(entry["finished"])
        if flags["compile_stopped_rays"]:
            entry["stopped"] = _mask(rays, proj["valid"] & (btype == STOP))
            history["stopped"].append(entry["stopped"])
        geo = SEGMENT_GEO if kind == "seg" else ARC_GEO
        optical = {f: v[proj["gather"][is_active]] for f, v in (opt_set or {}).items()
                   if f not in geo}
        optical["norm"] = proj["norm"][is_active]
        entry["optical"] = optical
        per_kind[kind] = entry

    kinds = [k for k in ("seg", "arc") if k in per_kind]
    for cls in ("active", "finished", "stopped"):
        sets = [per_kind[k][cls] for k in kinds if cls in per_kind[k]]
        if sets:
            result["rays"][cls] = amalgamate(sets) if len(sets) > 1 else sets[0]
    opt_order = kinds[::-1] if (bug_compatible and len(kinds) == 2) else kinds
    opts = [per_kind[k]["optical"] for k in opt_order]
    result["optical"] = amalgamate(opts) if len(opts) > 1 else opts[0]
    return result


def _react(system, proj, new_ray_length, index_type, finite_tir_gradient=False):
    """operation.py:255-307 StandardReaction.main."""
    rays = proj["rays"].get("active")
    if not rays or rays["x_start"].shape[0] == 0:
        return None
    if index_type == "index":
        mat_in = proj["optical"]["mat_in"].to(I64)
        mat_out = proj["optical"]["mat_out"].to(I64)
        wl = rays["wavelength"]
        n_stack = torch.stack([mat(wl) for mat in system.materials])
        rr = torch.arange(wl.shape[0], dtype=I64)
        n_in = n_stack[mat_in, rr]
        n_out = n_stack[mat_out, rr]
    else:
        n_in = proj["optical"]["n_in"]
        n_out = proj["optical"]["n_out"]
    new = {}
    if system.dimension == 2:
        new["x_start"], new["y_start"], new["x_end"], new["y_end"] = geom.snells_law_2D(
            rays["x_start"], rays["y_start"], rays["x_end"], rays["y_end"],
            proj["optical"]["norm"], n_in, n_out, new_ray_length,
            finite_tir_gradient=finite_tir_gradient)
    else:
        (new["x_start"], new["y_start"], new["z_start"],
         new["x_end"], new["y_end"], new["z_end"]) = geom.snells_law_3D(
            rays["x_start"], rays["y_start"], rays["z_start"],
            rays["x_end"], rays["y_end"], rays["z_end"],
            proj["optical"]["norm"], n_in, n_out, new_ray_length)
    return new


DEFAULT_FLAGS = dict(
    compile_dead_rays=False, compile_stopped_rays=False, compile_finished_rays=True,
    compile_active_rays=True, dead_ray_length=None,
)


def single_pass(system, rays, history, flags=None, new_ray_length=1.0,
                inherit=("wavelength",), index_type="index", chunk=2048, bug_compatible=False,
                finite_tir_gradient=False):
    """engine.py:2193-2302 with one StandardReaction operation.  Returns the new ray set
    (``{}`` when nothing reacted) and the projection result."""
    fl = dict(DEFAULT_FLAGS)
    fl.update(flags or {})
    rays = dict(rays)
    if system.dimension == 3:
        proj = _project_3d(system, rays, fl, history, chunk)
    else:
        proj = _project_2d(system, rays, fl, history, bug_compatible)
    new = _react(system, proj, new_ray_length, index_type, finite_tir_gradient)
    if new is None:
        return {}, proj
    for field in inherit:
        new[field] = proj["rays"]["active"][field]
    return new, proj


def ray_trace(system, sources, max_iterations=25, **kw):
    """engine.py:2311-2330.  Returns dict of amalgamated histories."""
    history = {"active": [], "finished": [], "stopped": [], "dead": []}
    rays = dict(sources)
    for _ in range(max_iterations):
        rays, _proj = single_pass(system, rays, history, **kw)
        if not rays:
            break
    out = {k: amalgamate(v) for k, v in history.items()}
    out["unfinished"] = rays
    return out
