"""
GPU parity of the boundary / constraint / vector-generator variants on the hot path that the
lens tests do not reach:

* ``MasterSlaveParametricTriangleBoundary``   (boundaries.py:1116-1229, gradient through the gather)
* ``ParametricCylindricalGuide`` parameter map (boundaries.py:1600-1617: repeat + cap padding + the
  vertex_update_map), rotationally symmetric and per-vertex
* ``ParametricSegmentBoundary`` / ``ParametricMultiSegmentBoundary`` (boundaries.py:528-826), 2-D
* ``PointConstraint`` / ``ClipConstraint`` (boundaries.py:124-158, 219-235) inside an update chain
* ``SecondSurfaceVG`` / ``FromPointVG`` / ``FromAxisVG`` (boundaries.py:239-383)
* ``OldestAncestor`` (operation.py:166-198)

Every scene is built through the product API; the oracle side restates the parameter -> vertex
map of the reference in plain float64 torch and traces with oracle/tracer.py; gradients come from
torch.autograd through the oracle.  Tolerances: float64 ray state, 1e-9 on rays, 1e-8 relative on
gradients.
"""
import math

import numpy as np
import pytest
import torch

from oracle import tracer

pytestmark = pytest.mark.gpu
PI = math.pi
GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
MATS = [tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]]


def _cpu(t):
    return t.detach().cpu()


def _tri_fields(verts, faces, update_map=None, mat=(1, 0)):
    f = tracer.faces_from_vertices(verts, faces, update_map)
    n = f["xp"].shape[0]
    if mat is not None:
        f["mat_in"] = torch.full((n,), mat[0], dtype=torch.int64)
        f["mat_out"] = torch.full((n,), mat[1], dtype=torch.int64)
    return f


def _oracle_target(target):
    return tracer.faces_from_vertices(_cpu(target._vertices), target._faces[:, 1:])


def _sources_for_oracle(system, keep=("wavelength",)):
    src = system._amalgamated_sources
    return {k: _cpu(src[k]).double() for k in GEO3 + tuple(keep)}


def _system3(optical, target, source):
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.operation as operation
    system = engine.OpticalSystem3D()
    system.optical = optical
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(3, [operation.StandardReaction()], ray_dtype=torch.float64,
                               simple_ray_inheritance={"wavelength"})
    eng.optical_system = system
    eng.validate_system()
    return system, eng


def _aperture_source(n, r_obj=0.2, r_ap=0.8):
    import tfrt.distributions as distributions
    import tfrt.drawing as drawing
    import tfrt.sources as sources
    start = distributions.StaticUniformCircle(n, r_obj)
    distributions.BasePointTransformation(start, translation=(-10, 0, 0))
    end = distributions.StaticUniformCircle(n, r_ap)
    distributions.BasePointTransformation(end)
    return sources.AperatureSource(3, start, end, [drawing.YELLOW], dense=False)


def _hex(k):
    import tfrt.mesh_tools as mt
    zp = mt.hexagonal_mesh(1.0, k)
    zp.rotate_y(90)
    zp.rotate_x(90)
    return zp


def _target_plane(x=10.0):
    import tfrt.boundaries as boundaries
    import tfrt.mesh_tools as mt
    t = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(x, 0, 0), direction=(1, 0, 0), i_size=100, j_size=100))
    t.frozen = True
    return t


def _spot_error(fin):
    return (fin["y_end"].double() ** 2 + fin["z_end"].double() ** 2).sum()


def _assert_grad(got, want, tol=1e-8):
    got, want = _cpu(got).double(), want.double()
    scale = float(want.abs().max())
    assert scale > 0, "oracle gradient is identically zero: the test exercises nothing"
    rel = float((got - want).abs().max()) / scale
    assert rel <= tol, f"gradient rel err {rel:.2e} > {tol}"


def _assert_rays(got, ref, fields=GEO3, tol=1e-9):
    assert got["x_start"].shape[0] == ref["x_start"].shape[0] > 0
    for f in fields:
        np.testing.assert_allclose(_cpu(got[f]).double().numpy(), ref[f].detach().numpy(),
                                   rtol=0, atol=tol, err_msg=f)


# ------------------------------------------------------------------------ master / slave

def test_master_slave_boundary_forward_and_gradient():
    """Mirror-symmetric front surface: the vertices with y >= 0 are masters, each vertex with
    y < 0 copies the parameter of its mirror image (boundaries.py:1116-1229).  The gradient of a
    master sums its own and its slave's vertex gradients (reverse of tf.gather, :1220)."""
    import tfrt.boundaries as boundaries
    zp = _hex(4)
    r2 = zp.points[:, 1] ** 2 + zp.points[:, 2] ** 2

    def filter_masters(verts):
        v = verts.detach().cpu().numpy() if isinstance(verts, torch.Tensor) else np.asarray(verts)
        return [int(i) for i in np.nonzero(v[:, 1] >= -1e-9)[0]]

    def attach_slaves(verts, master, available):
        v = verts.detach().cpu().numpy() if isinstance(verts, torch.Tensor) else np.asarray(verts)
        m = v[master]
        return {s for s in available
                if abs(v[s, 1] + m[1]) < 1e-9 and abs(v[s, 2] - m[2]) < 1e-9}

    front = boundaries.MasterSlaveParametricTriangleBoundary(
        filter_masters, attach_slaves, zp, boundaries.FromVectorVG((1, 0, 0)), flip_norm=True,
        initial_parameters=-(0.1 + 0.15 * (1 - r2)), material_dict={"mat_in": 1, "mat_out": 0})
    back = boundaries.ParametricTriangleBoundary(
        _hex(3), boundaries.FromVectorVG((1, 0, 0)), flip_norm=False,
        initial_parameters=0.12, material_dict={"mat_in": 1, "mat_out": 0})
    target = _target_plane()
    system, eng = _system3([front, back], target, _aperture_source(1500))
    n_v = front._zero_points.shape[0]
    n_masters = front.parameters.shape[0]
    assert n_masters < n_v and front._gather.shape[0] == n_v
    assert front["mat_in"].shape[0] == front["xp"].shape[0]

    eng.ray_trace(4)
    fin = eng.finished_rays
    err = _spot_error(fin)
    g_front, g_back = torch.autograd.grad(err, [front.parameters, back.parameters])

    q_f = _cpu(front.parameters).clone().requires_grad_(True)
    q_b = _cpu(back.parameters).clone().requires_grad_(True)
    gather = _cpu(front._gather).long()
    v_f = _cpu(front._zero_points) + q_f[gather].reshape(-1, 1) * _cpu(front._vectors)
    v_b = _cpu(back._zero_points) + q_b.reshape(-1, 1) * _cpu(back._vectors)
    optical = tracer.amalgamate([_tri_fields(v_f, front._faces[:, 1:]),
                                 _tri_fields(v_b, back._faces[:, 1:])])
    osys = tracer.System(3, materials=MATS, optical=optical, target=_oracle_target(target))
    ref = tracer.ray_trace(osys, _sources_for_oracle(system), max_iterations=4)
    _assert_rays(fin, ref["finished"])
    r_f, r_b = torch.autograd.grad(_spot_error(ref["finished"]), [q_f, q_b])
    assert fin["x_start"].shape[0] > 1200
    _assert_grad(g_front, r_f)
    _assert_grad(g_back, r_b)


# ------------------------------------------------------------------------ cylindrical guide

@pytest.mark.parametrize("symmetric", [True, False])
def test_cylindrical_guide_parameter_gradient(symmetric):
    """Gradient w.r.t. the guide's own parameters through ``repeat`` (rotationally symmetric),
    the cap padding and the vertex_update_map (boundaries.py:1600-1611)."""
    import tfrt.boundaries as boundaries
    import tfrt.mesh_tools as mt
    import tfrt.sources as sources
    theta_res, z_res = 12, 6
    guide = boundaries.ParametricCylindricalGuide(
        (0, 0, 0), (0, 0, 5), 0.5, theta_res=theta_res, z_res=z_res, initial_taper=(0.0, 0.2),
        rotationally_symmetric=symmetric, material_dict={"mat_in": 1, "mat_out": 0})
    assert guide.parameters.shape[0] == (z_res if symmetric else z_res * theta_res)
    if not symmetric:   # break the symmetry so that every parameter matters on its own
        with torch.no_grad():
            k = torch.arange(guide.parameters.shape[0], dtype=torch.float64,
                             device=guide.parameters.device)
            guide.parameters.add_(0.01 * torch.sin(1.7 * k))
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(0, 0, 4.9), direction=(0, 0, 1), i_size=4, j_size=4))
    rng = np.random.default_rng(11)
    n = 600
    src = sources.ManualSource(3)
    ang, tilt = rng.uniform(0, 2 * PI, n), rng.uniform(0.05, 0.4, n)
    r0 = rng.uniform(-0.2, 0.2, (n, 2))
    src["x_start"], src["y_start"], src["z_start"] = r0[:, 0], r0[:, 1], np.full(n, 0.2)
    src["x_end"] = r0[:, 0] + np.sin(tilt) * np.cos(ang)
    src["y_end"] = r0[:, 1] + np.sin(tilt) * np.sin(ang)
    src["z_end"] = 0.2 + np.cos(tilt)
    src["wavelength"] = np.full(n, 550.0)
    system, eng = _system3([guide], target, src)
    assert abs(float(guide.parameters.detach().min())) < 1e-15      # the constraint p -= min(p) ran (:1613)

    eng.ray_trace(10)
    fin = eng.finished_rays
    err = (fin["x_end"].double() ** 2 + fin["y_end"].double() ** 2).sum()
    (g,) = torch.autograd.grad(err, [guide.parameters])

    q = _cpu(guide.parameters).clone().requires_grad_(True)
    p = q.repeat_interleave(theta_res) if symmetric else q
    p = torch.cat([torch.zeros(1, dtype=torch.float64), p, torch.zeros(1, dtype=torch.float64)])
    verts = _cpu(guide._zero_points) + p.reshape(-1, 1) * _cpu(guide._vectors)
    np.testing.assert_allclose(verts.detach().numpy(), _cpu(guide.vertices).numpy(), atol=1e-15)
    assert guide.vertex_update_map is not None
    optical = _tri_fields(verts, guide.faces[:, 1:], guide.vertex_update_map)
    osys = tracer.System(3, materials=MATS, optical=optical, target=_oracle_target(target))
    ref = tracer.ray_trace(osys, _sources_for_oracle(system), max_iterations=10)
    assert fin["x_start"].shape[0] > 0.8 * n
    _assert_rays(fin, ref["finished"])
    rf = ref["finished"]
    (r,) = torch.autograd.grad((rf["x_end"] ** 2 + rf["y_end"] ** 2).sum(), [q])
    _assert_grad(g, r)


# --------------------------------------------------------------- constraints, vector generators

def test_point_and_clip_constraints_in_the_update_chain():
    """PointConstraint pins vertex 7 of the second surface 0.3 above the same vertex of the first
    (boundaries.py:124-158); a manually attached ClipConstraint clamps the first surface's
    parameters (boundaries.py:219-235).  The traced rays equal the oracle's on the constrained
    parameters."""
    import tfrt.boundaries as boundaries
    zp = _hex(3)
    r2 = zp.points[:, 1] ** 2 + zp.points[:, 2] ** 2
    lens = boundaries.ParametricMultiTriangleBoundary(
        zp, boundaries.FromVectorVG((1, 0, 0)),
        [boundaries.NoConstraint(), boundaries.PointConstraint(0.3, 7)], [True, False],
        initial_parameters=[-0.3 * (1 - r2), 0.25 * (1 - r2)],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2)
    s0, s1 = lens.surfaces
    s0.update_handles.append(boundaries.ClipConstraint(-0.2, -0.05).make(s0, None))
    raw0, raw1 = _cpu(s0.parameters).clone(), _cpu(s1.parameters).clone()
    target = _target_plane()
    system, eng = _system3(lens.surfaces, target, _aperture_source(800))
    want0 = raw0.clamp(-0.2, -0.05)
    want1 = raw1 + (want0[7] - raw1[7] + 0.3)
    assert float(raw0.min()) < -0.2 and float(raw0.max()) > -0.05    # the clip does something
    np.testing.assert_allclose(_cpu(s0.parameters).numpy(), want0.numpy(), atol=1e-15)
    np.testing.assert_allclose(_cpu(s1.parameters).numpy(), want1.numpy(), atol=1e-15)
    assert abs(float((s1.parameters[7] - s0.parameters[7]).detach()) - 0.3) < 1e-15
    system.update()                                                   # idempotent once satisfied
    np.testing.assert_allclose(_cpu(s1.parameters).numpy(), want1.numpy(), atol=1e-15)

    eng.ray_trace(4)
    surfs = []
    for s, p in zip(lens.surfaces, (want0, want1)):
        v = _cpu(s._zero_points) + p.reshape(-1, 1) * _cpu(s._vectors)
        surfs.append(_tri_fields(v, s._faces[:, 1:]))
    osys = tracer.System(3, materials=MATS, optical=tracer.amalgamate(surfs),
                         target=_oracle_target(target))
    ref = tracer.ray_trace(osys, _sources_for_oracle(system), max_iterations=4)
    _assert_rays(eng.finished_rays, ref["finished"])


@pytest.mark.parametrize("vg_kind", ["second_surface", "from_point", "from_axis"])
def test_vector_generators_forward_and_gradient(vg_kind):
    """The three vector generators the lens tests never use (boundaries.py:239-383): vectors equal
    a numpy restatement, and a surface moving along them traces and differentiates like the
    oracle."""
    import tfrt.boundaries as boundaries
    zp = _hex(3)
    zero = np.asarray(zp.points, dtype=np.float64)
    if vg_kind == "second_surface":
        second = zero + np.array([1.0, 0.0, 0.0]) + 0.2 * zero[:, [2, 1, 0]]
        vg = boundaries.SecondSurfaceVG(second)
        d = second - zero
    elif vg_kind == "from_point":
        vg = boundaries.FromPointVG((-4.0, 0.1, -0.05))
        d = zero - np.array([-4.0, 0.1, -0.05])
    else:
        # axis along z through (-3, 0, 0): vectors point away from the axis, perpendicular to it
        vg = boundaries.FromAxisVG((-3.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0))
        rel = zero - np.array([-3.0, 0.0, 0.0])
        d = rel - np.outer(rel @ np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0]))
    want = d / np.linalg.norm(d, axis=1, keepdims=True)
    r2 = zero[:, 1] ** 2 + zero[:, 2] ** 2
    front = boundaries.ParametricTriangleBoundary(
        zp, vg, flip_norm=True, initial_parameters=-(0.1 + 0.12 * (1 - r2)),
        material_dict={"mat_in": 1, "mat_out": 0})
    np.testing.assert_allclose(_cpu(front.vectors).numpy(), want, atol=1e-14)
    back = boundaries.ParametricTriangleBoundary(
        _hex(2), boundaries.FromVectorVG((1, 0, 0)), initial_parameters=0.15,
        material_dict={"mat_in": 1, "mat_out": 0})
    target = _target_plane()
    system, eng = _system3([front, back], target, _aperture_source(900, r_ap=0.7))
    eng.ray_trace(4)
    fin = eng.finished_rays
    (g,) = torch.autograd.grad(_spot_error(fin), [front.parameters])

    q = _cpu(front.parameters).clone().requires_grad_(True)
    v_f = _cpu(front._zero_points) + q.reshape(-1, 1) * torch.tensor(want)
    v_b = _cpu(back._zero_points) + _cpu(back.parameters).reshape(-1, 1) * _cpu(back._vectors)
    optical = tracer.amalgamate([_tri_fields(v_f, front._faces[:, 1:]),
                                 _tri_fields(v_b, back._faces[:, 1:])])
    osys = tracer.System(3, materials=MATS, optical=optical, target=_oracle_target(target))
    ref = tracer.ray_trace(osys, _sources_for_oracle(system), max_iterations=4)
    assert fin["x_start"].shape[0] > 600
    _assert_rays(fin, ref["finished"])
    (r,) = torch.autograd.grad(_spot_error(ref["finished"]), [q])
    _assert_grad(g, r)


def test_oldest_ancestor_is_inherited_to_every_generation():
    """operation.py:166-198: the source index tagged by ``annotate`` reaches the finished rays
    unchanged through every reaction."""
    import tfrt.boundaries as boundaries
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.operation as operation
    front = boundaries.ParametricTriangleBoundary(
        _hex(2), boundaries.FromVectorVG((1, 0, 0)), flip_norm=True, initial_parameters=-0.1,
        material_dict={"mat_in": 1, "mat_out": 0})
    back = boundaries.ParametricTriangleBoundary(
        _hex(2), boundaries.FromVectorVG((1, 0, 0)), initial_parameters=0.1,
        material_dict={"mat_in": 1, "mat_out": 0})
    system = engine.OpticalSystem3D()
    system.optical = [front, back]
    system.targets = [_target_plane()]
    system.sources = [_aperture_source(300, r_ap=0.6), _aperture_source(200, r_ap=0.5)]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    eng = engine.OpticalEngine(3, [operation.OldestAncestor(), operation.StandardReaction()],
                               ray_dtype=torch.float64)
    eng.optical_system = system
    system.update()
    eng.annotate()
    system.update()
    eng.validate_system()
    eng.ray_trace(4)
    fin = eng.finished_rays
    anc = fin["oldest_ancestor"].long()
    assert anc.shape[0] == 500
    assert torch.equal(anc, eng.last_trace["finished_id"].long())
    assert torch.equal(torch.sort(anc).values, torch.arange(500, device=anc.device))
    # the ancestor's source start point is where the pass-0 ray began: x = -10 for every ray
    src = system._amalgamated_sources
    assert torch.equal(src["oldest_ancestor"].long(), torch.arange(500, device=anc.device))


# ------------------------------------------------------------------------------------ 2-D

def _segments_from_points(points, flip):
    if flip:
        return points[1:, 0], points[1:, 1], points[:-1, 0], points[:-1, 1]
    return points[:-1, 0], points[:-1, 1], points[1:, 0], points[1:, 1]


def _trace2(system, eng, passes):
    eng.ray_trace(passes)
    return eng.finished_rays


def _system2(optical_segments, source, target_x=6.0):
    import tfrt.boundaries as boundaries
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.operation as operation
    target = boundaries.ManualSegmentBoundary()
    target.feed_segments(np.array([[target_x, -5.0, target_x, 5.0]]))
    system = engine.OpticalSystem2D()
    system.optical_segments = optical_segments
    system.target_segments = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(2, [operation.StandardReaction()], ray_dtype=torch.float64)
    eng.optical_system = system
    eng.validate_system()
    return system, eng, target


def _source2(n):
    import tfrt.sources as sources
    rng = np.random.default_rng(5)
    src = sources.ManualSource(2)
    y0 = rng.uniform(-0.8, 0.8, n)
    ang = rng.uniform(-0.05, 0.05, n)
    src["x_start"], src["y_start"] = np.full(n, -3.0), y0
    src["x_end"], src["y_end"] = -3.0 + np.cos(ang), y0 + np.sin(ang)
    src["wavelength"] = rng.uniform(450.0, 650.0, n)
    return src


def _oracle2(system, segs, target):
    tgt = {k: _cpu(target[k]).double() for k in ("x_start", "y_start", "x_end", "y_end")}
    osys = tracer.System(2, materials=MATS, optical_segments=segs, target_segments=tgt)
    src = {k: _cpu(v).double() for k, v in system._amalgamated_sources.items()
           if k in ("x_start", "y_start", "x_end", "y_end", "wavelength")}
    return osys, src


def test_parametric_segment_boundary_gradient():
    """boundaries.py:528-627: points = zero + p (one - zero), consecutive points form the segments
    (reversed by flip_norm).  Gradient of a spot error w.r.t. p against oracle autograd."""
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    k = 17
    ys = np.linspace(-1.2, 1.2, k)
    zero = distributions.ManualBasePointDistribution(2, points=np.stack([np.zeros(k), ys], 1))
    one = distributions.ManualBasePointDistribution(2, points=np.stack([np.ones(k), ys], 1))
    front = boundaries.ParametricSegmentBoundary(
        zero, one, flip_norm=True, initial_parameters=-(0.1 + 0.2 * (1 - (ys / 1.2) ** 2)),
        material_dict={"mat_in": 1, "mat_out": 0})
    back = boundaries.ParametricSegmentBoundary(
        zero, one, flip_norm=False, initial_parameters=0.1 + 0.15 * (1 - (ys / 1.2) ** 2),
        material_dict={"mat_in": 1, "mat_out": 0})
    system, eng, target = _system2([front, back], _source2(700))
    fin = _trace2(system, eng, 4)
    err = ((fin["y_end"].double() - 0.1) ** 2).sum()
    g_f, g_b = torch.autograd.grad(err, [front.parameters, back.parameters])

    q_f = _cpu(front.parameters).clone().requires_grad_(True)
    q_b = _cpu(back.parameters).clone().requires_grad_(True)
    z, o = _cpu(zero.points), _cpu(one.points)
    segs = []
    for q, flip in ((q_f, True), (q_b, False)):
        pts = z + q.reshape(-1, 1) * (o - z)
        xs, ys_, xe, ye = _segments_from_points(pts, flip)
        n = xs.shape[0]
        segs.append(dict(x_start=xs, y_start=ys_, x_end=xe, y_end=ye,
                         mat_in=torch.ones(n, dtype=torch.int64),
                         mat_out=torch.zeros(n, dtype=torch.int64)))
    osys, src = _oracle2(system, tracer.amalgamate(segs), target)
    ref = tracer.ray_trace(osys, src, max_iterations=4)
    rf = ref["finished"]
    assert fin["x_start"].shape[0] > 600
    _assert_rays(fin, rf, fields=("x_start", "y_start", "x_end", "y_end"))
    r_f, r_b = torch.autograd.grad(((rf["y_end"] - 0.1) ** 2).sum(), [q_f, q_b])
    _assert_grad(g_f, r_f)
    _assert_grad(g_b, r_b)


def test_parametric_multi_segment_boundary_with_constraints():
    """boundaries.py:631-826: two layers over shared base points, ThicknessConstraint between
    them; forward and gradient against the oracle on the constrained parameters."""
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    k = 13
    ys = np.linspace(-1.1, 1.1, k)
    zero = distributions.ManualBasePointDistribution(2, points=np.stack([np.zeros(k), ys], 1))
    one = distributions.ManualBasePointDistribution(2, points=np.stack([np.ones(k), ys], 1))
    bump = 1 - (ys / 1.1) ** 2
    multi = boundaries.ParametricMultiSegmentBoundary(
        zero, one,
        [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.15, "min")],
        [True, False], initial_parameters=[-0.2 * bump - 0.05, 0.2 * bump],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2)
    system, eng, target = _system2([multi], _source2(500))
    p0, p1 = [_cpu(p) for p in multi.parameters]
    assert abs(float(p0.min())) < 1e-15                      # p0 += max(0 - p0) + 0 (:208-215)
    assert abs(float((p1 - p0).min()) - 0.15) < 1e-15
    fin = _trace2(system, eng, 4)
    err = (fin["y_end"].double() ** 2).sum()
    g0, g1 = torch.autograd.grad(err, multi.parameters)

    q0, q1 = p0.clone().requires_grad_(True), p1.clone().requires_grad_(True)
    z, o = _cpu(zero.points), _cpu(one.points)
    segs = []
    for q, flip in ((q0, True), (q1, False)):
        pts = z + q.reshape(-1, 1) * (o - z)
        xs, ys_, xe, ye = _segments_from_points(pts, flip)
        n = xs.shape[0]
        segs.append(dict(x_start=xs, y_start=ys_, x_end=xe, y_end=ye,
                         mat_in=torch.ones(n, dtype=torch.int64),
                         mat_out=torch.zeros(n, dtype=torch.int64)))
    osys, src = _oracle2(system, tracer.amalgamate(segs), target)
    ref = tracer.ray_trace(osys, src, max_iterations=4)
    rf = ref["finished"]
    assert fin["x_start"].shape[0] > 400
    _assert_rays(fin, rf, fields=("x_start", "y_start", "x_end", "y_end"))
    r0, r1 = torch.autograd.grad((rf["y_end"] ** 2).sum(), [q0, q1])
    _assert_grad(g0, r0)
    _assert_grad(g1, r1)
