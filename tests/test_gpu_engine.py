"""
GPU parity of the tfrt-style Python API (engine / boundaries / sources / optimizer) against
the oracle: the same scene is built through the product API and, from the product objects'
raw arrays, through the oracle.
"""
import math

import numpy as np
import pytest
import torch

import oracle_util
from oracle import tracer

pytestmark = pytest.mark.gpu
PI = math.pi


def _t(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def _oracle_surface(surface, params):
    zero = surface._zero_points.detach().cpu()
    vec = surface._vectors.detach().cpu()
    verts = zero + params.reshape(-1, 1) * vec
    f = tracer.faces_from_vertices(verts, surface._faces[:, 1:], surface.vertex_update_map)
    n = f["xp"].shape[0]
    f["mat_in"] = torch.ones(n, dtype=torch.int64)
    f["mat_out"] = torch.zeros(n, dtype=torch.int64)
    return f


def _build_lens(n_rays, k=4, ray_dtype=torch.float64, random_rays=False, **engine_kw):
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.drawing as drawing
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources

    circle = distributions.RandomUniformCircle if random_rays else distributions.StaticUniformCircle
    start_points = circle(n_rays, 0.2)
    distributions.BasePointTransformation(start_points, translation=(-10, 0, 0))
    end_points = circle(n_rays, 0.8)
    distributions.BasePointTransformation(end_points)
    source = sources.AperatureSource(
        3, start_points, end_points, [drawing.YELLOW], dense=False,
        extra_fields={"object_coords": ("start_point", start_points, "points")})

    zero_points = mt.hexagonal_mesh(1.0, k)
    zero_points.rotate_y(90)
    zero_points.rotate_x(90)
    r2 = (zero_points.points[:, 1] ** 2 + zero_points.points[:, 2] ** 2)
    rng = np.random.default_rng(0)
    vmap = rng.uniform(size=(zero_points.n_faces, 3)) > 0.25
    lens = boundaries.ParametricMultiTriangleBoundary(
        zero_points, boundaries.FromVectorVG((1, 0, 0)),
        [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.2, "min")],
        [True, False],
        initial_parameters=[-0.15 * (1 - r2), 0.15 * (1 - r2)],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2,
        vertex_update_map=vmap)
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(10, 0, 0), direction=(1, 0, 0), i_size=100, j_size=100))
    target.frozen = True

    system = engine.OpticalSystem3D()
    system.optical = lens.surfaces
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(
        3, [operation.StandardReaction()],
        simple_ray_inheritance={"wavelength", "object_coords"}, ray_dtype=ray_dtype, **engine_kw)
    eng.optical_system = system
    eng.validate_system()
    return eng, system, lens, target, source


def _oracle_for(system, lens, target, source, params):
    surfs = [_oracle_surface(s, p) for s, p in zip(lens.surfaces, params)]
    optical = tracer.amalgamate(surfs)
    tgt = tracer.faces_from_vertices(target._vertices.detach().cpu(), target._faces[:, 1:])
    osys = tracer.System(3, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
                         optical=optical, target=tgt)
    src = {k: v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu()
           for k, v in system._amalgamated_sources.items()}
    return osys, src


def test_engine_forward_matches_oracle():
    eng, system, lens, target, source = _build_lens(2000)
    eng.ray_trace(4)
    params = [p.detach().cpu().clone() for p in lens.parameters]
    osys, src = _oracle_for(system, lens, target, source, params)
    ref = tracer.ray_trace(osys, src, max_iterations=4, inherit=("wavelength", "object_coords"))
    fin = eng.finished_rays
    assert set(fin.keys()) == set(ref["finished"].keys())
    assert fin["x_start"].shape[0] == ref["finished"]["x_start"].shape[0] > 1500
    for f in ref["finished"].keys():
        np.testing.assert_allclose(fin[f].detach().cpu().double().numpy(),
                                   ref["finished"][f].numpy(), rtol=0, atol=1e-9)
    act = eng.active_rays
    for f in ("x_start", "y_end", "z_end", "wavelength"):
        np.testing.assert_allclose(act[f].detach().cpu().double().numpy(),
                                   ref["active"][f].numpy(), rtol=0, atol=1e-9)
    # constraints ran (boundaries.py:208-215): p0 -= min(p0); min(p1 - p0) == 0.2
    p0, p1 = [p.detach().cpu() for p in lens.parameters]
    assert abs(float(p0.min())) < 1e-12
    assert abs(float((p1 - p0).min()) - 0.2) < 1e-12


def test_optimizer_step_matches_oracle_autograd():
    import tfrt.optimizer as optimizer
    eng, system, lens, target, source = _build_lens(1500, k=3, ray_dtype=torch.float32)

    def error_function(engine):
        fin = engine.finished_rays
        out = torch.stack([fin["y_end"], fin["z_end"]], dim=1).double()
        goal = -fin["object_coords"][:, 1:]
        return ((out - goal) ** 2)

    opt = optimizer.SGD_Optimizer(eng, lens.parameters, error_function, 3, learning_rate=1.0,
                                  grad_clip=1e9)
    before = [p.detach().cpu().clone() for p in lens.parameters]
    grads, err_sum, n_terms = opt.raw_gradient()
    # oracle: constraints are applied by update() before the trace, so read the parameters
    # the trace actually used
    used = [p.detach().cpu().clone() for p in lens.parameters]
    q = [u.clone().requires_grad_(True) for u in used]
    osys, src = _oracle_for(system, lens, target, source, q)
    for k in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"):
        src[k] = src[k].float().double()  # float32 ray state on the GPU side
    ref = tracer.ray_trace(osys, src, max_iterations=3, inherit=("wavelength", "object_coords"))
    rf = ref["finished"]
    rerr = ((torch.stack([rf["y_end"], rf["z_end"]], 1) + rf["object_coords"][:, 1:]) ** 2)
    rg = torch.autograd.grad(rerr.sum(), q)
    assert n_terms == rerr.numel()
    assert abs(float(err_sum) - float(rerr.sum().detach())) <= 1e-5 * float(rerr.sum().detach())
    for g, r in zip(grads, rg):
        rel = (g.cpu() - r).abs().max() / r.abs().max()
        assert rel < 1e-5, f"gradient rel err {rel:.2e}"

    # one full step: p <- p - 0.01 * clip(lr * g)
    err = opt.single_step(None)
    assert np.isfinite(err)
    moved = sum(float((p.detach().cpu() - b).abs().sum()) for p, b in zip(lens.parameters, before))
    assert moved > 0


def test_single_pass_result_structure():
    eng, system, lens, target, source = _build_lens(500, k=2)
    rays = dict(system._amalgamated_sources)
    new = eng.single_pass(rays)
    res = eng.last_projection_result
    assert set(res["rays"].keys()) >= {"active"}
    n_act = res["rays"]["active"]["x_start"].shape[0]
    assert res["optical"]["mat_in"].shape[0] == n_act
    assert res["optical"]["norm"].shape == (n_act, 3)
    assert new["x_start"].shape[0] == n_act
    assert set(new.keys()) == {"x_start", "y_start", "z_start", "x_end", "y_end", "z_end",
                               "wavelength", "object_coords"}
    # new rays start where the projected active rays end
    np.testing.assert_allclose(new["x_start"].detach().cpu().numpy(),
                               res["rays"]["active"]["x_end"].detach().cpu().numpy(), atol=1e-12)


def test_cylindrical_guide_total_internal_reflection():
    """Light guide (dev/light_guide.py style, 3-D): rays launched inside an acrylic cylinder
    bounce by TIR off the wall until they reach the end cap; compare with the oracle."""
    import tfrt.boundaries as boundaries
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.operation as operation
    import tfrt.sources as sources
    import tfrt.mesh_tools as mt

    guide = boundaries.ParametricCylindricalGuide(
        (0, 0, 0), (0, 0, 6), 0.5, theta_res=16, z_res=7, initial_taper=(0.0, 0.15),
        material_dict={"mat_in": 1, "mat_out": 0})
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(0, 0, 5.9), direction=(0, 0, 1), i_size=3, j_size=3))
    rng = np.random.default_rng(3)
    n = 800
    src = sources.ManualSource(3)
    ang = rng.uniform(0, 2 * math.pi, n)
    tilt = rng.uniform(0.05, 0.35, n)
    r0 = rng.uniform(0, 0.3, (n, 2))
    src["x_start"], src["y_start"], src["z_start"] = r0[:, 0], r0[:, 1], np.full(n, 0.2)
    src["x_end"] = r0[:, 0] + np.sin(tilt) * np.cos(ang)
    src["y_end"] = r0[:, 1] + np.sin(tilt) * np.sin(ang)
    src["z_end"] = 0.2 + np.cos(tilt)
    src["wavelength"] = np.full(n, 550.0)
    system = engine.OpticalSystem3D()
    system.optical = [guide]
    system.targets = [target]
    system.sources = [src]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(3, [operation.StandardReaction()], ray_dtype=torch.float64)
    eng.optical_system = system
    eng.ray_trace(12)
    fin = eng.finished_rays
    assert fin["x_start"].shape[0] > 0.9 * n                      # guided to the end
    assert eng.last_trace["counts"][:, 1].nonzero()[0].size >= 2  # after different bounce counts

    verts = guide.vertices.detach().cpu()
    opt = tracer.faces_from_vertices(verts, guide.faces[:, 1:])
    nf = opt["xp"].shape[0]
    opt["mat_in"] = torch.ones(nf, dtype=torch.int64)
    opt["mat_out"] = torch.zeros(nf, dtype=torch.int64)
    tgt = tracer.faces_from_vertices(target.vertices.detach().cpu(), target.faces[:, 1:])
    osys = tracer.System(3, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
                         optical=opt, target=tgt)
    osrc = {k: v.detach().cpu().double() for k, v in system._amalgamated_sources.items()}
    ref = tracer.ray_trace(osys, osrc, max_iterations=12)
    for f in ("x_start", "z_start", "x_end", "y_end", "z_end"):
        np.testing.assert_allclose(fin[f].detach().cpu().numpy(), ref["finished"][f].numpy(), atol=1e-9)


def test_ghost_through_goes_straight():
    import tfrt.operation as operation
    eng, system, lens, target, source = _build_lens(500, k=2)
    import tfrt.engine as engine
    ghost = engine.OpticalEngine(3, [operation.GhostThrough()], ray_dtype=torch.float64)
    ghost.optical_system = system
    ghost.ray_trace(4)
    fin = ghost.finished_rays
    src = system._amalgamated_sources
    # straight lines from the source through the lens to the target plane x = 10
    d = torch.stack([src[a + "_end"] - src[a + "_start"] for a in "xyz"], 1)
    t = (10.0 - src["x_start"]) / d[:, 0]
    want_y = src["y_start"] + t * d[:, 1]
    ids = ghost.last_trace["finished_id"].long()
    np.testing.assert_allclose(fin["y_end"].detach().cpu().numpy(), want_y[ids].cpu().numpy(), atol=1e-9)


def test_stops_and_technical_intersections_through_the_engine():
    """A stop plate in front of the lens: stopped rays are compiled, and with
    compile_technical_intersections the stop / target boundary data is gathered to the rays
    that hit them (engine.py:2135-2191); classes and end points equal the oracle's."""
    import tfrt.boundaries as boundaries
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    eng0, system0, lens, target, source = _build_lens(3000, k=3, ray_dtype=torch.float64)
    stop = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(-1.0, 0.25, 0.0), direction=(1, 0, 0), i_size=0.3, j_size=0.3))
    system = engine.OpticalSystem3D()
    system.optical = lens.surfaces
    system.stops = [stop]
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(
        3, [operation.StandardReaction()], compile_stopped_rays=True, compile_dead_rays=True,
        compile_technical_intersections=True, simple_ray_inheritance={"wavelength"},
        ray_dtype=torch.float64)
    eng.optical_system = system
    eng.validate_system()

    eng.ray_trace(4)
    params = [p.detach().cpu().clone() for p in lens.parameters]
    surfs = [_oracle_surface(s, p) for s, p in zip(lens.surfaces, params)]
    faces_of = lambda b: tracer.faces_from_vertices(b._vertices.detach().cpu(), b._faces[:, 1:])
    osys = tracer.System(3, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
                         optical=tracer.amalgamate(surfs), stop=faces_of(stop),
                         target=faces_of(target))
    src = {k: v.detach().cpu().double() for k, v in system._amalgamated_sources.items()
           if k in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength")}
    ref = tracer.ray_trace(osys, src, max_iterations=4, inherit=("wavelength",),
                           flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    n_stopped = ref["stopped"]["x_start"].shape[0]
    assert 20 < n_stopped < 1500
    for name, got in (("stopped", eng.stopped_rays), ("finished", eng.finished_rays)):
        for f in ("x_start", "y_end", "z_end"):
            np.testing.assert_allclose(got[f].detach().cpu().numpy(), ref[name][f].numpy(),
                                       rtol=0, atol=1e-9, err_msg=f"{name}.{f}")
    # technical boundary data of the first pass: stop norm for every stopped ray
    eng.clear_ray_history()
    eng.single_pass(dict(system._amalgamated_sources))
    res = eng.last_projection_result
    assert res["rays"]["stopped"]["x_start"].shape[0] == n_stopped
    assert set(res["stop"].keys()) >= {"xp", "x1", "z2", "norm"}
    assert res["stop"]["norm"].shape == (n_stopped, 3)
    np.testing.assert_allclose(res["stop"]["norm"].detach().abs().cpu().numpy()[:, 0], 1.0, atol=1e-12)
    assert "target" not in res or res["target"]["xp"].shape[0] == 0   # nothing reaches it in pass 1


def test_custom_operation_main_runs_the_reference_pass_loop():
    """A user reaction written against the plug-in API (operation.py:25-160: ``main(engine,
    proj_result)`` returns ``{"active": {"rays", "valid"}}``): a mirror implemented in torch on
    the projection result.  The engine then runs the reference's Python pass loop
    (engine.py:2193-2330) around the fused projection.  Result = the oracle's StandardReaction on
    the same scene with a reflective material (n_in = 0 mirrors, geometry.py:745-747)."""
    import tfrt.boundaries as boundaries
    import tfrt.engine as engine
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources

    class Mirror(operation.RayOperation):
        calls = 0

        @property
        def optical_signature(self):
            return set()

        def main(self, eng, proj):
            Mirror.calls += 1
            act = proj["rays"].get("active")
            if act is None or act["x_start"].shape[0] == 0:
                return {}
            s = torch.stack([act["x_start"], act["y_start"], act["z_start"]], 1).double()
            h = torch.stack([act["x_end"], act["y_end"], act["z_end"]], 1).double()
            n = proj["optical"]["norm"]
            u = (h - s) / torch.linalg.norm(h - s, dim=1, keepdim=True)
            w = u - 2.0 * (u * n).sum(1, keepdim=True) * n
            e = h + eng.new_ray_length * w
            rays = {"x_start": h[:, 0], "y_start": h[:, 1], "z_start": h[:, 2],
                    "x_end": e[:, 0], "y_end": e[:, 1], "z_end": e[:, 2]}
            return {"active": {"rays": rays,
                               "valid": torch.ones(h.shape[0], dtype=torch.bool, device=h.device)}}

    # a tilted mirror facet field (hex mesh, bumpy) in front of a target wall the light returns to
    zp = mt.hexagonal_mesh(1.0, 3)
    zp.rotate_y(90)
    zp.rotate_x(90)
    r2 = zp.points[:, 1] ** 2 + zp.points[:, 2] ** 2
    mirror = boundaries.ParametricTriangleBoundary(
        zp, boundaries.FromVectorVG((1, 0, 0)), initial_parameters=0.2 * r2)
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(-4, 0, 0), direction=(1, 0, 0), i_size=30, j_size=30))
    rng = np.random.default_rng(2)
    n = 700
    src = sources.ManualSource(3)
    y0, z0 = rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n)
    src["x_start"], src["y_start"], src["z_start"] = np.full(n, -2.0), y0, z0
    src["x_end"], src["y_end"], src["z_end"] = np.full(n, -1.0), y0 + 0.02, z0 - 0.01
    src["wavelength"] = np.full(n, 600.0)
    src["tag"] = np.arange(n, dtype=np.float64)
    system = engine.OpticalSystem3D()
    system.optical = [mirror]
    system.targets = [target]
    system.sources = [src]
    system.update()
    eng = engine.OpticalEngine(3, [Mirror()], ray_dtype=torch.float64,
                               simple_ray_inheritance={"tag"})
    eng.optical_system = system
    eng.ray_trace(4)
    assert Mirror.calls >= 2                               # the operation's Python code did run
    fin = eng.finished_rays
    assert fin["x_start"].shape[0] > 600

    opt = tracer.faces_from_vertices(mirror.vertices.detach().cpu(), mirror.faces[:, 1:])
    nf = opt["xp"].shape[0]
    opt["mat_in"] = torch.zeros(nf, dtype=torch.int64)      # material 0 = reflective (n = 0)
    opt["mat_out"] = torch.zeros(nf, dtype=torch.int64)
    tgt = tracer.faces_from_vertices(target.vertices.detach().cpu(), target.faces[:, 1:])
    osys = tracer.System(3, materials=[tracer.MATERIALS["reflective"]], optical=opt, target=tgt)
    osrc = {k: v.detach().cpu().double() for k, v in system._amalgamated_sources.items()}
    ref = tracer.ray_trace(osys, osrc, max_iterations=4, inherit=("wavelength", "tag"))
    rf = ref["finished"]
    assert fin["x_start"].shape[0] == rf["x_start"].shape[0]
    for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "tag"):
        np.testing.assert_allclose(fin[f].detach().cpu().numpy(), rf[f].numpy(), rtol=0, atol=1e-9,
                                   err_msg=f)
    # differentiable through the Python reaction: d(spot size)/d(mirror parameters) vs the oracle
    loss = (fin["y_end"] ** 2 + fin["z_end"] ** 2).sum()
    (g,) = torch.autograd.grad(loss, [mirror.parameters])
    q = mirror.parameters.detach().cpu().clone().requires_grad_(True)
    verts = mirror._zero_points.detach().cpu() + q.reshape(-1, 1) * mirror._vectors.detach().cpu()
    opt2 = tracer.faces_from_vertices(verts, mirror.faces[:, 1:])
    opt2["mat_in"], opt2["mat_out"] = opt["mat_in"], opt["mat_out"]
    osys2 = tracer.System(3, materials=[tracer.MATERIALS["reflective"]], optical=opt2, target=tgt)
    rf2 = tracer.ray_trace(osys2, osrc, max_iterations=4, inherit=("wavelength", "tag"))["finished"]
    (r,) = torch.autograd.grad((rf2["y_end"] ** 2 + rf2["z_end"] ** 2).sum(), [q])
    rel = float((g.cpu() - r).abs().max() / r.abs().max())
    assert rel < 1e-8, f"gradient rel err {rel:.2e}"
