"""
Pins the parts of the oracle the reference holds no tests for (triangles, Snell, nearest hit,
trace loop, gradients) with analytic optics and finite differences.
"""
import math

import numpy as np
import torch

from oracle import geom, tracer

PI = math.pi
t = lambda *a: torch.tensor(a, dtype=torch.float64)


def test_config1_known_answer():
    """dev/single_pass.py scene: hit x and refracted angles from SURVEY.md section 8c."""
    arc = dict(x_center=t(5.), y_center=t(0.), angle_start=t(3 * PI / 4), angle_end=t(5 * PI / 4),
               radius=t(5.), mat_in=torch.tensor([1]), mat_out=torch.tensor([0]))
    system = tracer.System(2, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
                           optical_arcs=arc)
    y = torch.linspace(-1.5, 1.5, 10, dtype=torch.float64)
    src = dict(x_start=-torch.ones_like(y), y_start=y, x_end=torch.zeros_like(y), y_end=y,
               wavelength=680 * torch.ones_like(y))
    hist = {"active": [], "finished": [], "stopped": [], "dead": []}
    new, proj = tracer.single_pass(system, src, hist)
    np.testing.assert_allclose(proj["rays"]["active"]["x_end"].numpy()[:5],
                               [0.2303, 0.1380, 0.0699, 0.0251, 0.0028], atol=5e-5)
    ang = torch.atan2(new["y_end"] - new["y_start"], new["x_end"] - new["x_start"]).numpy()
    np.testing.assert_allclose(ang[:5], [0.10188, 0.07819, 0.05531, 0.03297, 0.01096], atol=2e-5)
    assert abs(float(tracer.MATERIALS["acrylic"](t(680.))) - 1.4893789) < 1e-6


def _plate(n_glass=1.5):
    """Two parallel planes x=0 (norm -x) and x=1 (norm +x), glass between, target at x=3."""
    def quad(x, flip):
        v = torch.tensor([[x, -50., -50.], [x, 50., -50.], [x, 50., 50.], [x, -50., 50.]], dtype=torch.float64)
        f = [[0, 2, 1], [0, 3, 2]] if flip else [[0, 1, 2], [0, 2, 3]]
        return tracer.faces_from_vertices(v, f)
    front, back, target = quad(0.0, True), quad(1.0, False), quad(3.0, False)
    assert float(front["norm"][0, 0]) == -1.0 and float(back["norm"][0, 0]) == 1.0
    for s in (front, back):
        s["n_in"] = torch.full((2,), n_glass, dtype=torch.float64)
        s["n_out"] = torch.ones(2, dtype=torch.float64)
    return tracer.System(3, optical=tracer.amalgamate([front, back]), target=target)


def test_plane_parallel_plate_exit_parallel_to_entry():
    system = _plate()
    d = torch.tensor([[1.0, 0.3, -0.2], [1.0, -0.5, 0.1], [1.0, 0.0, 0.0]], dtype=torch.float64)
    s = torch.tensor([[-1.0, 0.1, 0.2]] * 3, dtype=torch.float64)
    src = {f"{a}_start": s[:, i] for i, a in enumerate("xyz")}
    src.update({f"{a}_end": (s + d)[:, i] for i, a in enumerate("xyz")})
    out = tracer.ray_trace(system, src, 5, inherit=(), index_type="value")
    fin = out["finished"]
    assert fin["x_start"].shape[0] == 3
    dout = torch.stack([fin[f"{a}_end"] - fin[f"{a}_start"] for a in "xyz"], 1)
    dout = dout / dout.norm(dim=1, keepdim=True)
    din = d / d.norm(dim=1, keepdim=True)
    np.testing.assert_allclose(dout.numpy(), din.numpy(), atol=1e-12)
    # Snell inside the glass: sin(theta_in) = 1.5 sin(theta_glass)
    act = out["active"]
    inner = torch.stack([act[f"{a}_end"][3:] - act[f"{a}_start"][3:] for a in "xyz"], 1)
    inner = inner / inner.norm(dim=1, keepdim=True)
    sin_in = torch.sqrt(1 - din[:, 0] ** 2)
    sin_gl = torch.sqrt(1 - inner[:, 0] ** 2)
    np.testing.assert_allclose(sin_in.numpy(), 1.5 * sin_gl.numpy(), atol=1e-12)


def test_total_internal_reflection_threshold_and_mirror_law():
    n = 1.5
    crit = math.asin(1 / n)
    norm = torch.tensor([[1.0, 0.0, 0.0]] * 3, dtype=torch.float64)  # glass on -x side (n_in), air +x
    for theta, expect_tir in ((crit - 1e-3, False), (crit + 1e-3, True)):
        s = t(-math.cos(theta), -math.sin(theta), 0.0).reshape(1, 3)
        o = geom.snells_law_3D(s[:, 0], s[:, 1], s[:, 2], t(0.), t(0.), t(0.), norm[:1],
                               t(n), t(1.0), 1.0)
        w = torch.stack(o[3:], 1)[0]
        if expect_tir:
            np.testing.assert_allclose(w.numpy(), [-math.cos(theta), math.sin(theta), 0.0], atol=1e-12)
        else:
            assert float(w[0]) > 0  # transmitted
            np.testing.assert_allclose(float(w[1]), n * math.sin(theta), atol=1e-12)
    # mirror: n_in == 0 always reflects (geometry.py:747): w = u - 2 (n.u) n
    u = torch.tensor([[0.6, 0.8, 0.0]], dtype=torch.float64)
    o = geom.snells_law_3D(-u[:, 0], -u[:, 1], -u[:, 2], t(0.), t(0.), t(0.), norm[:1], t(0.0), t(1.0), 2.0)
    np.testing.assert_allclose(torch.stack(o[3:], 1)[0].numpy(), [-1.2, 1.6, 0.0], atol=1e-12)


def test_snell_2d_equals_3d_for_refraction():
    rng = np.random.default_rng(0)
    n = 500
    s = rng.normal(size=(n, 2)); h = rng.normal(size=(n, 2))
    na = rng.uniform(-PI, PI, n)
    n_in = rng.uniform(1.0, 1.7, n); n_out = rng.uniform(1.0, 1.7, n)
    o2 = geom.snells_law_2D(s[:, 0], s[:, 1], h[:, 0], h[:, 1], na, n_in, n_out, 1.3)
    norm3 = np.stack([np.cos(na), np.sin(na), np.zeros(n)], 1)
    z = np.zeros(n)
    o3 = geom.snells_law_3D(s[:, 0], s[:, 1], z, h[:, 0], h[:, 1], z, norm3, n_in, n_out, 1.3)
    np.testing.assert_allclose(o2[2].numpy(), o3[3].numpy(), atol=1e-12)
    np.testing.assert_allclose(o2[3].numpy(), o3[4].numpy(), atol=1e-12)


def test_nearest_hit_first_index_wins_ties_and_dense_equals_chunked():
    rng = np.random.default_rng(4)
    m, n = 40, 300
    tri = rng.normal(size=(m, 9))
    tri[7] = tri[3]  # duplicate triangle: tf.argmin keeps the lower index
    rays = rng.normal(size=(6, n)) * 2
    args = [torch.tensor(rays[i]) for i in range(6)] + [torch.tensor(tri[:, i]) for i in range(9)]
    a = tracer.intersection_3d(*args, 1e-10, 1e-10, 1e-10, chunk=10 ** 6)
    b = tracer.intersection_3d(*args, 1e-10, 1e-10, 1e-10, chunk=37)
    hit = a[3]
    assert torch.equal(hit, b[3])
    for x, y in zip(a, b):  # invalid entries are garbage (sentinel = 2*max over the chunk)
        assert torch.equal(x[hit], y[hit])
    assert int(hit.sum()) > 20
    assert not bool((a[8][hit] == 7).any())


def test_oracle_gradient_matches_finite_differences():
    import scene_util
    import oracle_util
    scene = scene_util.lens_scene(60, k_front=2, k_back=2, seed=7)

    def loss(p_f, p_b):
        system, (q_f, q_b), _ = oracle_util.lens_oracle(scene, p_f, p_b)
        ref = tracer.ray_trace(system, oracle_util.source_dict(scene["rays"], scene["wavelength"]),
                               max_iterations=4, inherit=("wavelength", "ray_id"))
        fin = ref["finished"]
        return (fin["y_end"] ** 2 + fin["z_end"] ** 2).sum(), q_f, q_b

    val, q_f, q_b = loss(scene["p_f"], scene["p_b"])
    g_f, g_b = torch.autograd.grad(val, [q_f, q_b])
    eps = 1e-6
    for which, g in (("p_f", g_f), ("p_b", g_b)):
        for k in (0, 5, 11):
            pp = {n: scene[n].copy() for n in ("p_f", "p_b")}
            pp[which][k] += eps
            up = float(loss(pp["p_f"], pp["p_b"])[0])
            pp[which][k] -= 2 * eps
            dn = float(loss(pp["p_f"], pp["p_b"])[0])
            fd = (up - dn) / (2 * eps)
            assert abs(fd - float(g[k])) <= 1e-5 * max(1.0, abs(fd)), (which, k, fd, float(g[k]))
