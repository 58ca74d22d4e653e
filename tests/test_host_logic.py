"""
Host-side logic of the API mirror (no GPU): update plumbing, constraints, sources and
distributions, mesh tools, engine field assembly / inheritance, optimizer gradient
processing.  Where a HIP op would be needed the ``cpu_backend`` fixture swaps in the oracle.
"""
import math

import numpy as np
import pytest
import torch

PI = math.pi


def test_update_order_and_frozen():
    from tensorflowraytrace_amd.update import RecursivelyUpdatable
    log = []

    class Thing(RecursivelyUpdatable):
        def _update(self):
            log.append("self")

        def _generate_update_handles(self):
            return [lambda: log.append("child")]

    t = Thing()
    t.post_update_handles.append(lambda: log.append("post"))
    assert log == ["child", "self"]  # constructor updates once (update.py:50)
    log.clear()
    t.update()
    assert log == ["child", "self", "post"]
    t.frozen = True
    log.clear()
    t.update()
    assert log == []
    t.forced_update()
    assert log == ["child", "self"]  # forced_update skips post handles, like the reference
    t.frozen = False
    t.recursively_update = False
    log.clear()
    t.update()
    assert log == ["self", "post"]


def test_variable_semantics():
    from tensorflowraytrace_amd.variable import Variable
    v = Variable([1.0, 2.0, 3.0], device="cpu")
    assert v.dtype == torch.float64 and v.requires_grad and v.is_leaf
    v.assign_add(1.0)
    v.assign_sub([0.5, 0.5, 0.5])
    np.testing.assert_allclose(v.numpy(), [1.5, 2.5, 3.5])
    v.assign([0.0, 1.0, 2.0])
    (v * v).sum().backward()
    np.testing.assert_allclose(v.grad.numpy(), [0.0, 2.0, 4.0])


def test_hexagonal_mesh_counts_and_orientation():
    import tensorflowraytrace_amd.mesh_tools as mt
    for k in (1, 3, 9):
        m = mt.hexagonal_mesh(1.0, k)
        assert m.n_points == 3 * k * k + 3 * k + 1 and m.n_faces == 6 * k * k
        tri = m.points[m.triangles()]
        area2 = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])[:, 2]
        assert np.all(area2 > 0)                       # counter-clockwise
        np.testing.assert_allclose(area2, area2[0])    # equilateral, equal area
        assert len({tuple(sorted(f)) for f in m.triangles()}) == m.n_faces
    m = mt.hexagonal_mesh(2.0, 2)
    np.testing.assert_allclose(m.points[0], 0)          # centre first, then ring by ring
    np.testing.assert_allclose(np.linalg.norm(m.points[1:7], axis=1), 1.0)
    np.testing.assert_allclose(m.points[7], [2.0, 0, 0], atol=1e-15)
    m.rotate_y(90).rotate_x(90)
    assert np.abs(m.points[:, 0]).max() < 1e-15         # now in the y-z plane


def test_stl_round_trip(tmp_path):
    import tensorflowraytrace_amd.mesh_tools as mt
    m = mt.hexagonal_mesh(1.0, 2)
    f = tmp_path / "m.stl"
    m.save(str(f))
    r = mt.read(str(f))
    assert r.n_faces == m.n_faces and r.n_points == m.n_points
    a = np.sort(m.points[m.triangles()].reshape(m.n_faces, -1), axis=0)
    b = np.sort(r.points[r.triangles()].reshape(r.n_faces, -1), axis=0)
    np.testing.assert_allclose(a, b)


def test_distributions_and_sources_dense_undense(cpu_backend):
    import tensorflowraytrace_amd.distributions as distributions
    import tensorflowraytrace_amd.sources as sources
    c = distributions.StaticUniformCircle(5, 2.0)
    idx = np.arange(5) + 0.5
    np.testing.assert_allclose(c.points[:, 0].numpy(),
                               2.0 * np.sqrt(idx / 5) * np.cos(PI * (1 + 5 ** 0.5) * idx))
    beam = distributions.StaticUniformBeam(-1.5, 1.5, 10)
    np.testing.assert_allclose(beam.points[:, 0].numpy(), 0, atol=1e-15)
    np.testing.assert_allclose(beam.points[:, 1].numpy(), np.linspace(-1.5, 1.5, 10))
    angles = distributions.StaticUniformAngularDistribution(-0.1, 0.1, 3)
    src = sources.AngularSource(2, (-1.0, 0.0), 0.0, angles, beam, [680.0, 450.0])
    n = 3 * 10 * 2
    assert all(src[f].shape == (n,) for f in ("x_start", "y_start", "x_end", "y_end", "wavelength"))
    np.testing.assert_allclose(src["x_start"].numpy(), -1.0)
    # dense = every combination exactly once
    combos = {(round(float(a), 6), round(float(b), 6), float(w)) for a, b, w in zip(
        torch.atan2(src["y_end"] - src["y_start"], src["x_end"] - src["x_start"]), src["y_start"],
        src["wavelength"])}
    assert len(combos) == n
    # undense aperture source with an extra field lifted to 3-D by BasePointTransformation
    a = distributions.StaticUniformCircle(7, 0.2)
    distributions.BasePointTransformation(a, translation=(-10, 0, 0))
    b = distributions.StaticUniformCircle(7, 0.9)
    distributions.BasePointTransformation(b)
    ap = sources.AperatureSource(3, a, b, [575.0], dense=False,
                                 extra_fields={"object_coords": ("start_point", a, "points")})
    assert ap["x_start"].shape == (7,) and ap["object_coords"].shape == (7, 3)
    np.testing.assert_allclose(ap["x_start"].numpy(), -10.0)
    np.testing.assert_allclose(ap["z_end"].numpy(), b.points[:, 2].numpy())
    with pytest.raises(ValueError):
        sources.AperatureSource(3, a, distributions.StaticUniformCircle(5, 1.0), [575.0], dense=False)


def test_quaternion_helpers():
    import tensorflowraytrace_amd.distributions as d
    q = d.get_rotation_quaternion_from_u_to_v((1.0, 0, 0), (0, 1.0, 0))
    v = d.rotate_vector_by_quaternion(q, torch.tensor([[1.0, 0, 0], [0, 0, 1.0]], dtype=torch.float64))
    np.testing.assert_allclose(v.numpy(), [[0, 1, 0], [0, 0, 1]], atol=1e-15)
    q = d.quaternion_from_euler((PI / 2, 0, 0))
    v = d.rotate_vector_by_quaternion(q, torch.tensor([[0, 1.0, 0]], dtype=torch.float64))
    np.testing.assert_allclose(v.numpy(), [[0, 0, 1]], atol=1e-15)


def _lens_api(n_rays=300, k=2, end_radius=0.8, target_size=100):
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources
    a = distributions.StaticUniformCircle(n_rays, 0.2)
    distributions.BasePointTransformation(a, translation=(-10, 0, 0))
    b = distributions.StaticUniformCircle(n_rays, end_radius)
    distributions.BasePointTransformation(b)
    source = sources.AperatureSource(3, a, b, [575.0], dense=False,
                                     extra_fields={"object_coords": ("start_point", a, "points")})
    zp = mt.hexagonal_mesh(1.0, k)
    zp.rotate_y(90)
    zp.rotate_x(90)
    r2 = zp.points[:, 1] ** 2 + zp.points[:, 2] ** 2
    lens = boundaries.ParametricMultiTriangleBoundary(
        zp, boundaries.FromVectorVG((1, 0, 0)),
        [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.2, "min")],
        [True, False], initial_parameters=[-0.15 * (1 - r2), 0.15 * (1 - r2)],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2)
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(10, 0, 0), direction=(1, 0, 0), i_size=target_size, j_size=target_size))
    system = engine.OpticalSystem3D()
    system.optical = lens.surfaces
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(3, [operation.StandardReaction()],
                               simple_ray_inheritance={"wavelength", "object_coords"},
                               ray_dtype=torch.float64)
    eng.optical_system = system
    eng.validate_system()
    return eng, system, lens, target


def test_boundaries_constraints_and_merge(cpu_backend):
    eng, system, lens, target = _lens_api()
    p0, p1 = lens.parameters
    assert abs(float(p0.min())) < 1e-14                    # ThicknessConstraint(0, 'min') vs zero
    assert abs(float((p1 - p0).min()) - 0.2) < 1e-14       # 0.2 'min' vs previous surface
    s0, s1 = lens.surfaces
    assert float(s0["norm"][:, 0].max()) < 0 < float(s1["norm"][:, 0].min())  # flip_norm
    assert system._optical_count == 48 and system._target_count == 2
    m = system._merged
    assert m["xp"].shape == (50,) and m["norm"].shape == (50, 3)
    assert m["catagory"].tolist() == [0] * 48 + [2] * 2      # optical, stop, target order
    assert set(lens.keys()) >= {"xp", "norm", "mat_in"} and lens["xp"].shape == (48,)
    np.testing.assert_allclose(s0["xp"].detach().numpy(),
                               s0.face_verts[:, 0].detach().numpy())
    with pytest.raises(RuntimeError):
        s0.update_from_mesh()


def test_engine_outputs_and_inheritance(cpu_backend):
    from oracle import tracer
    eng, system, lens, target = _lens_api()
    eng.ray_trace(4)
    fin = eng.finished_rays
    n = fin["x_start"].shape[0]
    assert n > 200
    assert set(fin.keys()) == {"x_start", "y_start", "z_start", "x_end", "y_end", "z_end",
                               "wavelength", "object_coords"}
    assert fin["object_coords"].shape == (n, 3)
    np.testing.assert_allclose(fin["x_end"].detach().numpy(), 10.0, atol=1e-12)
    # inherited field really belongs to the ray's source ancestor: finished ray starts on the
    # back surface, which the source ray through object_coords also reaches
    act = eng.active_rays
    assert act["x_start"].shape[0] >= 2 * n - 5
    assert bool(eng.all_rays)
    eng.clear_ray_history()
    assert not bool(eng.finished_rays)
    # single_pass keeps the reference's result structure
    new = eng.single_pass(dict(system._amalgamated_sources))
    res = eng.last_projection_result
    assert "active" in res["rays"] and "optical" in res
    assert new["x_start"].shape == res["rays"]["active"]["x_end"].shape


def test_optimizer_gradient_processing(cpu_backend):
    import tfrt.optimizer as optimizer
    eng, system, lens, target = _lens_api(200)

    def erf(engine):
        fin = engine.finished_rays
        out = torch.stack([fin["y_end"], fin["z_end"]], 1)
        return (out + fin["object_coords"][:, 1:]) ** 2

    opt = optimizer.SGD_Optimizer(eng, lens.parameters, erf, 3, learning_rate=2.0,
                                  individual_lr=[1.0, 0.5], grad_clip=1e-3)
    raw, err_sum, n_terms = opt.raw_gradient()
    assert n_terms > 0 and float(err_sum) > 0
    acc = [torch.eye(raw[0].numel(), dtype=torch.float64) * 2.0, None]
    proc, mean = opt.process_gradient(acc, lr_scale=0.5)
    want0 = torch.clamp(raw[0] * (0.5 * 1.0 * 2.0), -1e-3, 1e-3) * 2.0
    want1 = torch.clamp(raw[1] * (0.5 * 0.5 * 2.0), -1e-3, 1e-3)
    np.testing.assert_allclose(proc[0].numpy(), want0.numpy(), rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(proc[1].numpy(), want1.numpy(), rtol=1e-12, atol=1e-18)
    assert abs(float(mean) - float(err_sum) / n_terms) < 1e-15
    before = [p.detach().clone() for p in lens.parameters]
    e = opt.single_step(None, lr_scale=1.0)
    assert np.isfinite(e) and opt.iterations == 1
    delta = lens.parameters[1].detach() - before[1]
    assert float(delta.abs().max()) <= 0.01 * 1e-3 + 1e-15   # Keras SGD lr 0.01 x clipped grad
    sm = torch.full((raw[0].numel(), raw[0].numel()), 1.0 / raw[0].numel(), dtype=torch.float64)
    opt.smooth(lens.parameters[0], sm)
    assert float(lens.parameters[0].detach().std()) < 1e-15
    with pytest.raises(ValueError):
        opt.momentum = 1.5
    with pytest.raises(ValueError):
        optimizer.SGD_Optimizer(eng, lens.parameters[0], erf, 3)


def test_shard_bounds_cover_exactly():
    from tensorflowraytrace_amd.distributed import shard_bounds
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_mesh_tools_parametrisation_and_smoothing():
    import tensorflowraytrace_amd.mesh_tools as mt
    h = mt.hexagonal_mesh(1.0, 3)
    top = mt.get_closest_point(h, (0, 0, 0))
    assert top == 0
    vmap, acc = mt.mesh_parametrization_tools(h, top)
    assert vmap.shape == (h.n_faces, 3) and vmap.dtype == bool and vmap.any(axis=1).all()
    tri = h.triangles()
    r = np.linalg.norm(h.points, axis=1)
    # a face never moves its innermost vertex (the sweep front that reached it)
    for f in range(h.n_faces):
        inner = tri[f][np.argmin(r[tri[f]])]
        assert not vmap[f][list(tri[f]).index(inner)] or np.isclose(r[tri[f]], r[inner]).all()
    n = h.n_points
    assert acc.shape == (n, n) and np.allclose(np.diag(acc), 1.0)
    assert acc[0].sum() == 1.0                     # the top parent has no ancestors
    assert np.all(acc[:, 0] == 1.0)                # and is everybody's ancestor
    outer = np.argmax(r)
    assert acc[outer].sum() > 3                    # ancestors chain back to the centre
    sm = mt.mesh_smoothing_tool(h, [4, 2, 1])
    assert np.allclose(sm.sum(axis=1), 1.0) and np.isclose(sm[0, 0], 4 / 7)
    assert np.isclose(sm[0, 1], 2 / 7 / 6)         # six first neighbours share 2/7
    sub = mt.mesh_smoothing_tool(h, [1, 1], active_vertices=range(7))
    assert sub.shape == (7, 7)
    c = mt.circular_mesh(1.0, 0.2)
    t = c.points[c.triangles()]
    area = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])[:, 2]
    assert (area > 0).all() and abs(area.sum() / 2 - math.pi) < 0.05
    w = mt.circular_mesh(1.0, 0.2, starting_radius=0.5, theta_start=0, theta_end=math.pi / 2)
    t = w.points[w.triangles()]
    area = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])[:, 2]
    assert (area > 0).all() and abs(area.sum() / 2 - math.pi / 4 * 0.75) < 0.02
    cyl = mt.cylindrical_mesh((0, 0, 0), (0, 0, 2), 0.5, theta_res=8, z_res=5)
    assert cyl.n_points == 8 * 5 + 2 and cyl.n_faces == 8 * 2 + 8 * 2 * 4
    edges = {}
    for f in cyl.triangles():
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            edges[(a, b)] = edges.get((a, b), 0) + 1
    assert all(v == 1 for v in edges.values()) and all((b, a) in edges for a, b in edges)  # closed, oriented


def test_cylindrical_guide(cpu_backend):
    import tensorflowraytrace_amd.boundaries as boundaries
    g = boundaries.ParametricCylindricalGuide((0, 0, 0), (0, 0, 4), 0.5, theta_res=6, z_res=5,
                                              initial_taper=(0.0, 0.4),
                                              material_dict={"mat_in": 1, "mat_out": 0})
    assert g.parameters.shape == (30,) and abs(float(g.parameters.min())) < 1e-15
    v = g.vertices.detach().numpy()
    rad = np.hypot(v[1:-1, 0], v[1:-1, 1]).reshape(5, 6)
    np.testing.assert_allclose(rad, 0.5 + np.linspace(0, 0.4, 5)[:, None] * np.ones((1, 6)), atol=1e-12)
    np.testing.assert_allclose(v[0], [0, 0, 0]); np.testing.assert_allclose(v[-1], [0, 0, 4])
    assert g["xp"].shape == (g.faces.shape[0],) and g["mat_in"].shape == g["xp"].shape
    assert g.accumulator.shape == (32, 32)
    sym = boundaries.ParametricCylindricalGuide((0, 0, 0), (0, 0, 4), 0.5, theta_res=6, z_res=5,
                                                rotationally_symmetric=True, initial_parameters=0.1)
    assert sym.parameters.shape == (5,)
    loss = (sym["xp"] ** 2).sum()
    gp, = torch.autograd.grad(loss, [sym.parameters])
    assert gp.shape == (5,) and float(gp.abs().sum()) > 0


def test_stl_binary_and_ascii_round_trip(tmp_path):
    """boundary.save / file_name= (boundaries.py:859-874 via pyvista): binary STL by default,
    ASCII on request; topology survives the trip (vertices merged exactly)."""
    import tfrt.mesh_tools as mt
    mesh = mt.hexagonal_mesh(1.0, 3)
    mesh.rotate_y(30)
    for binary in (True, False):
        path = str(tmp_path / f"mesh_{binary}.stl")
        mesh.save(path, binary=binary)
        back = mt.read(path)
        assert (back.n_points, back.n_faces) == (mesh.n_points, mesh.n_faces)
        tol = 1e-6 if binary else 0.0                     # binary STL stores float32
        assert np.abs(back.points[back.triangles()] - mesh.points[mesh.triangles()]).max() <= tol
    import os
    assert os.path.getsize(str(tmp_path / "mesh_True.stl")) == 84 + 50 * mesh.n_faces
    with open(str(tmp_path / "bad.stl"), "w") as f:
        f.write("solid x\nvertex 0 0 0\nvertex 1 0 0\nendsolid x\n")
    with pytest.raises(ValueError):
        mt.read(str(tmp_path / "bad.stl"))


def test_sphere_mesh_is_closed_and_outward():
    import tfrt.mesh_tools as mt
    from collections import Counter
    m = mt.sphere(0.5, (1.0, -2.0, 0.25), theta_resolution=9, phi_resolution=7)
    assert m.n_points == 2 + 9 * 5 and m.n_faces == 2 * 9 + 2 * 9 * 4
    np.testing.assert_allclose(np.linalg.norm(m.points - [1.0, -2.0, 0.25], axis=1), 0.5, rtol=1e-12)
    edges = Counter()
    for a, b, c in m.triangles():
        for e in ((a, b), (b, c), (c, a)):
            edges[frozenset(e)] += 1
    assert set(edges.values()) == {2}                     # watertight
    tri = m.points[m.triangles()]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert (np.sum(n * (tri.mean(1) - [1.0, -2.0, 0.25]), axis=1) > 0).all()
    with pytest.raises(ValueError):
        mt.sphere(1.0, theta_resolution=2)


def test_cluster_order_is_a_permutation_with_compact_aligned_runs():
    """The numpy k-d face ordering (the yardstick of tfrt_cluster_order, tests/test_gpu_order.py):
    a permutation whose aligned runs of 16 and 128 faces are spatially compact (k-d subtrees)."""
    import types
    import cluster_reference
    ops = types.SimpleNamespace(cluster_order=cluster_reference.cluster_order_numpy)
    import tfrt.mesh_tools as mt
    mesh = mt.hexagonal_mesh(1.0, 20)                     # 2400 faces
    tri = torch.tensor(mesh.points[mesh.triangles()].reshape(-1, 9))
    order = ops.cluster_order(tri).numpy()
    assert np.array_equal(np.sort(order), np.arange(tri.shape[0]))
    cent = tri.reshape(-1, 3, 3).mean(1).numpy()[order]

    def mean_radius(run):
        n = cent.shape[0] // run * run
        c = cent[:n].reshape(-1, run, 3)
        return np.linalg.norm(c - c.mean(1, keepdims=True), axis=2).max(1).mean()

    edge = 1.0 / 20
    assert mean_radius(16) < 4 * edge                     # ~4x4 face patches
    assert mean_radius(128) < 10 * edge
    shuffled = np.random.default_rng(0).permutation(cent.shape[0])
    cent_shuffled = cent[shuffled]
    c = cent_shuffled[:2400 // 16 * 16].reshape(-1, 16, 3)
    assert mean_radius(16) < 0.2 * np.linalg.norm(c - c.mean(1, keepdims=True), axis=2).max(1).mean()
    # tiny inputs
    assert ops.cluster_order(tri[:5]).numel() == 5 and ops.cluster_order(tri[:0]).numel() == 0


def test_deferred_scalar_behaves_like_the_float_the_reference_returns():
    from tensorflowraytrace_amd.optimizer import DeferredScalar
    d = DeferredScalar(torch.tensor(2.5, dtype=torch.float64))
    assert float(d) == 2.5 and d == 2.5 and d < 3 and d >= 2.5 and abs(-d) == 2.5
    assert d + 1 == 3.5 and 1 + d == 3.5 and d * 2 == 5.0 and 5 / d == 2.0 and d - 0.5 == 2.0
    assert f"{d:.2f}" == "2.50" and repr(d) == "2.5" and np.isfinite(d) and bool(d)
    assert np.asarray(d).dtype == np.float64 and d.numpy() == 2.5 and d.item() == 2.5
    assert np.mean([d, DeferredScalar(torch.tensor(3.5))]) == 3.0


def test_engine_trace_mode_selection():
    import tfrt.engine as engine
    import tfrt.operation as operation

    class Sys:
        def __init__(self, m):
            self._merged_face_verts = torch.zeros((m, 9))

    e3 = engine.OpticalEngine(3, [operation.StandardReaction()])
    assert e3._trace_mode(Sys(10)) == "all-pairs" and e3._trace_mode(Sys(64)) == "group"
    for setting, want in ((False, "all-pairs"), ("all-pairs", "all-pairs"), (True, "group"),
                          ("sort", "group"), ("group", "group")):
        e = engine.OpticalEngine(3, [operation.StandardReaction()], accelerate=setting)
        assert e._trace_mode(Sys(5000)) == want
    e2 = engine.OpticalEngine(2, [operation.StandardReaction()], accelerate="group")
    assert e2._trace_mode(Sys(5000)) == "all-pairs"       # 2-D has its own filter
    with pytest.raises(ValueError):
        engine.OpticalEngine(3, [operation.StandardReaction()], accelerate="fastest")


# ------------------------------------------------------------------------------------------
# mesh helpers (tfrt/mesh_tools.py:28-217, 956-1160)

def test_pack_unpack_faces_round_trip():
    import tensorflowraytrace_amd.mesh_tools as mt
    m = mt.hexagonal_mesh(1.0, 3)
    tri = mt.unpack_faces(m.faces)
    np.testing.assert_array_equal(tri, m.triangles())
    np.testing.assert_array_equal(mt.pack_faces(tri), m.faces)
    assert mt.pack_faces(np.zeros((0, 3), dtype=np.int64)).shape == (0,)


def test_gaussian_weights():
    import tensorflowraytrace_amd.mesh_tools as mt
    w = mt.gaussian_weights(2.0, 4)
    np.testing.assert_allclose(w, np.exp(-0.5 * (np.arange(4) / 2.0) ** 2), rtol=0, atol=0)
    assert w[0] == 1.0 and np.all(np.diff(w) < 0)


def test_neighbour_helpers_and_generations_of_hexagonal_mesh():
    import tensorflowraytrace_amd.mesh_tools as mt
    m = mt.hexagonal_mesh(1.0, 4)
    edges = mt.get_unique_edges_1p(m)
    # Euler: V - E + F = 1 for a disc
    assert m.n_points - len(edges) + m.n_faces == 1
    faces = mt.get_faces_as_sets(m)
    for v in (0, 5, m.n_points - 1):
        assert mt.neighbors_from_edges(v, edges) == mt.neighbors_from_faces(v, faces)
    assert len(mt.neighbors_from_edges(0, edges)) == 6           # hexagon centre
    gens = mt.find_generations(0, m)
    assert [len(g) for g in gens] == [1, 6, 12, 18, 24]           # rings of 6k vertices
    assert set().union(*gens) == set(range(m.n_points))
    r = np.linalg.norm(m.points, axis=1)
    for k, g in enumerate(gens):                                  # ring k sits at hexagonal radius k/4
        assert np.all(r[list(g)] <= k / 4 + 1e-12)


def test_single_parent_relationships_form_a_spanning_tree():
    import tensorflowraytrace_amd.mesh_tools as mt
    m = mt.hexagonal_mesh(1.0, 3)
    acc, rel = mt.gradient_accumulator_1p(m, origin=(0, 0, 0))
    top = rel["top_parent"]
    assert top == 0
    n = m.n_points
    assert all(len(p) == 1 for i, p in enumerate(rel["parent"]) if i != top)
    assert rel["parent"][top] == set() and rel["descendant"][top] == set(range(n)) - {top}
    assert sum(len(c) for c in rel["child"]) == n - 1
    for v in range(n):                                            # ancestor/descendant are inverse relations
        for a in rel["ancestor"][v]:
            assert v in rel["descendant"][a]
    # accumulator = identity + descendants; the root row sums everything
    assert acc.shape == (n, n) and np.all(acc[top] == 1)
    assert acc.sum() == n + sum(len(d) for d in rel["descendant"])


def test_clean_mesh_merges_vertices_and_drops_bad_faces():
    import tensorflowraytrace_amd.mesh_tools as mt
    m = mt.hexagonal_mesh(1.0, 3)
    tri = m.triangles()
    nv = m.n_points
    # five near-copies of existing vertices, used by some faces; duplicate + degenerate faces
    pts = np.concatenate([m.points, m.points[:5] + 2e-4])
    dirty = tri.copy()
    for k in range(5):
        rows = np.nonzero((dirty == k).any(axis=1))[0][:2]
        sub = dirty[rows]
        sub[sub == k] = nv + k
        dirty[rows] = sub
    dirty = np.concatenate([dirty, dirty[:4][:, [1, 2, 0]], [[0, 0, 1]], [[2, nv + 2, 3]]])
    v, f = mt.clean_mesh_raw(pts, dirty.copy(), distance_tolerance=1e-6)   # (2e-4)^2*3 < 1e-6
    assert v.shape == (nv, 3)
    np.testing.assert_array_equal(v, m.points)
    np.testing.assert_array_equal(f, tri)                          # order and winding preserved
    # squared-distance semantics of the reference: the same offsets survive a tighter tolerance
    v2, f2 = mt.clean_mesh_raw(pts, dirty.copy(), distance_tolerance=1e-8)
    assert v2.shape[0] == nv + 5
    cleaned = mt.clean_mesh(mt.PolyData(pts, mt.pack_faces(dirty)), 1e-6)
    assert cleaned.n_points == nv and cleaned.n_faces == m.n_faces


def test_planar_interpolated_remesh_reproduces_a_plane_and_a_paraboloid():
    import tensorflowraytrace_amd.mesh_tools as mt
    src = mt.hexagonal_mesh(1.0, 12)
    src.points[:, 2] = 0.3 * src.points[:, 0] - 0.2 * src.points[:, 1] + 0.1
    base = mt.hexagonal_mesh(0.8, 5)
    flat, h = mt.planar_interpolated_remesh(src, base)
    np.testing.assert_allclose(h, 0.3 * base.points[:, 0] - 0.2 * base.points[:, 1] + 0.1, atol=1e-12)
    assert np.all(flat.points[:, 2] == 0) and np.all(base.points[:, 2] == 0)
    bowl = mt.hexagonal_mesh(1.0, 24)
    bowl.points = bowl.points[:, [2, 0, 1]]                                # into the y-z plane
    bowl.points[:, 0] = bowl.points[:, 1] ** 2 + bowl.points[:, 2] ** 2    # height along x
    base_x = mt.hexagonal_mesh(0.8, 5)
    base_x.points = base_x.points[:, [2, 0, 1]]
    out = mt.planar_interpolated_remesh(bowl, base_x, range_axis=0, flatten=False)
    want = out.points[:, 1] ** 2 + out.points[:, 2] ** 2
    assert np.abs(out.points[:, 0] - want).max() < 5e-3
    far = mt.PolyData(np.array([[5.0, 5.0, 0.0]]), np.zeros(0, dtype=np.int64))
    _, fill = mt.planar_interpolated_remesh(src, far, interp_fill_value=-7.0)
    assert fill[0] == -7.0
    with pytest.raises(ValueError):
        mt.planar_interpolated_remesh(src, base, range_axis=3)


# ------------------------------------------------------------------------------------------
# analysis helpers (tfrt/analyze.py)

def test_histogram2d_matches_numpy_and_clips_outliers_into_edge_bins():
    import tfrt.analyze as analyze
    rng = np.random.default_rng(5)
    x, y = rng.uniform(-1, 1, 5000), rng.uniform(-2, 2, 5000)
    H = analyze.histogram2D(torch.tensor(x), torch.tensor(y), ((-1, 1), (-2, 2)), x_bins=8, y_bins=5)
    assert H.shape == (5, 8) and H.dtype == torch.int32           # y is the first index
    want, _, _ = np.histogram2d(y, x, bins=(5, 8), range=((-2, 2), (-1, 1)))
    np.testing.assert_array_equal(H.numpy(), want.astype(np.int32))
    # tf.histogram_fixed_width semantics: out-of-range points are counted in the edge bins
    H2 = analyze.histogram2D(torch.tensor([-9.0, 9.0, 0.1]), torch.tensor([0.0, 9.0, -9.0]),
                             ((-1, 1), (-1, 1)), x_bins=4)
    assert H2.shape == (4, 4) and int(H2.sum()) == 3
    assert H2[2, 0] == 1 and H2[3, 3] == 1 and H2[0, 2] == 1
    assert analyze.histogram2D(torch.zeros(0), torch.zeros(0), ((0, 1), (0, 1)), 3).sum() == 0


def test_inner_product_and_imaging_test_without_display():
    import tfrt.analyze as analyze
    a = np.array([[1.0, 2.0], [3.0, 4.0]])
    assert abs(analyze.inner_product(a, 3 * a) - 1.0) < 1e-15
    assert abs(analyze.inner_product([[1.0, 0.0]], [[0.0, 1.0]])) < 1e-15
    rng = np.random.default_rng(1)
    calls = []

    def get_samples():
        calls.append(1)
        return torch.tensor(rng.uniform(-1, 1, (100, 2)))

    h, xe, ye, image = analyze.imaging_test(get_samples, ((-1, 1), (-1, 1)), batch_count=4, bins=16,
                                            verbose=False, display=False)
    assert len(calls) == 4 and h.shape == (16, 16) and h.sum() == 400 and image is None
    assert xe.shape == (17,) and ye[0] == -1 and ye[-1] == 1


def test_distribution_differential_prefers_the_goal_distribution():
    import tfrt.analyze as analyze
    rng = np.random.default_rng(2)
    goal = lambda gx, gy: torch.exp(-(gx ** 2 + gy ** 2) / (2 * 0.3 ** 2))     # noqa: E731
    dd = analyze.DistributionDifferential(goal, ((-1, 1), (-1, 1)), x_bins=20)
    good = rng.normal(0, 0.3, (2, 40000))
    flat = rng.uniform(-1, 1, (2, 40000))
    q_good = float(dd(torch.tensor(good[0]), torch.tensor(good[1])))
    assert dd.saved_histo.shape == (20, 20)
    q_flat = float(dd(torch.tensor(flat[0]), torch.tensor(flat[1])))
    assert 0 <= q_good < 0.02 < q_flat
    # array goal + out-of-bounds penalty: mean of penalty(distance to the domain centre)
    arr = torch.ones(6, 6, dtype=torch.float64)
    pen = analyze.DistributionDifferential(arr, ((0, 2), (0, 2)), oob_penalty=lambda d: 2.0 * d)
    inside = rng.uniform(0, 2, (2, 20000))
    x = torch.tensor(np.concatenate([inside[0], [5.0, 1.0]]))
    y = torch.tensor(np.concatenate([inside[1], [1.0, -3.0]]))
    base = analyze.DistributionDifferential(arr, ((0, 2), (0, 2)))
    q0 = float(base(torch.tensor(inside[0]), torch.tensor(inside[1])))
    assert abs(float(pen(x, y)) - (q0 + 2.0 * 4.0)) < 1e-12           # both strays are 4 away
    assert abs(float(pen(torch.tensor(inside[0]), torch.tensor(inside[1]))) - q0) < 1e-12
    with pytest.raises(ValueError):
        analyze.DistributionDifferential(torch.ones(3), ((0, 1), (0, 1)))
    with pytest.raises(ValueError):
        analyze.DistributionDifferential(arr, (0, 1))
    with pytest.raises(ValueError):
        analyze.DistributionDifferential(arr, ((0, 1), (0, 1)), oob_penalty=lambda d: d.nope)
