"""
csrc/trace_math*.h (the float64 per-ray math the HIP kernels execute) compiled for the host
and checked against the oracle: exact-stage decisions bit for bit, Snell to rounding, and the
hand-derived adjoints against torch.autograd through the oracle.
"""
import ctypes
import math

import numpy as np
import torch

from oracle import geom, tracer

P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
D, I = ctypes.c_double, ctypes.c_int64
PI = math.pi


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _hit_case(rng, n):
    P9 = rng.normal(size=(n, 3, 3))
    bary = rng.dirichlet([1, 1, 1], size=n)
    hit = np.einsum("nc,ncd->nd", bary, P9)
    s = hit + rng.normal(size=(n, 3)) * 2
    e = s + (hit - s) * rng.uniform(0.2, 3.0, size=(n, 1))
    return s, e, P9.reshape(n, 9)


def test_exact_triangle_bitwise_and_edge_cases(host_math):
    rng = np.random.default_rng(0)
    n = 4000
    s, e, P9 = _hit_case(rng, n)
    # add misses, back-facing starts, degenerate rays and parallel rays
    s[:500] = rng.normal(size=(500, 3)) * 3
    e[500:520] = s[500:520]
    P9[520:540, 6:9] = P9[520:540, 3:6]  # zero-area triangles
    ru, tu, tv = np.zeros(n), np.zeros(n), np.zeros(n)
    va, hh = np.zeros(n, np.uint8), np.zeros((n, 3))
    host_math.hm_exact_triangle(I(n), P(s), P(e), P(P9), D(1e-10), D(1e-10), D(1e-10),
                                P(ru), P(tu), P(tv), P(va), P(hh))
    x, y, z, valid, ray_u, trig_u, trig_v = geom.raw_line_triangle_intersect(
        *[s[:, i] for i in range(3)], *[e[:, i] for i in range(3)], *[P9[:, i] for i in range(9)], 1e-10)
    valid = valid & (trig_u >= -1e-10) & (trig_v >= -1e-10) & (trig_u + trig_v <= 1 + 1e-10) & (ray_u >= 1e-10)
    assert np.array_equal(va.astype(bool), valid.numpy())
    assert np.array_equal(ru, ray_u.numpy()) and np.array_equal(tu, trig_u.numpy())
    assert np.array_equal(hh[:, 0], x.numpy())
    assert 0 < va.sum() < n


def test_snell3d_and_adjoint3d(host_math):
    rng = np.random.default_rng(1)
    n = 3000
    L = 1.7
    s, e, P9 = _hit_case(rng, n)
    n_in = rng.choice([1.0, 1.5, 0.0, 1.33], size=n)
    n_out = rng.choice([1.0, 1.5, 1.2], size=n)
    child = rng.integers(0, 2, size=n).astype(np.uint8)
    st, et, Pt = [torch.tensor(a, requires_grad=True) for a in (s, e, P9)]
    x, y, z, valid, ru, tu, tv = geom.raw_line_triangle_intersect(
        *[st[:, i] for i in range(3)], *[et[:, i] for i in range(3)], *[Pt[:, i] for i in range(9)], 1e-10)
    h = torch.stack([x, y, z], 1)
    norm = tracer.faces_from_vertices(Pt.reshape(-1, 3), np.arange(3 * n).reshape(n, 3))["norm"]
    o = geom.snells_law_3D(st[:, 0], st[:, 1], st[:, 2], x, y, z, norm, torch.tensor(n_in),
                           torch.tensor(n_out), L)
    cs, ce = torch.stack(o[:3], 1), torch.stack(o[3:], 1)
    en = np.zeros((n, 3))
    hh = h.detach().numpy().copy()
    host_math.hm_snell3d(I(n), P(s), P(hh), P(P9), P(n_in), P(n_out), D(L), P(en))
    assert np.abs(en - ce.detach().numpy()).max() < 1e-12
    g = [rng.normal(size=(n, 3)) for _ in range(4)]
    cm = torch.tensor(child.astype(np.float64)).reshape(-1, 1)
    loss = (st * torch.tensor(g[0])).sum() + (h * torch.tensor(g[1])).sum() + \
        (cm * (cs * torch.tensor(g[2]) + ce * torch.tensor(g[3]))).sum()
    gs_r, ge_r, gP_r = torch.autograd.grad(loss, [st, et, Pt])
    gs, ge, gP = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 9))
    ruu = ru.detach().numpy().copy()
    gh = g[1] + child[:, None] * g[2]
    host_math.hm_adjoint3d(I(n), P(s), P(e), P(P9), P(ruu), P(child), P(n_in), P(n_out), D(L),
                           P(g[0]), P(gh), P(g[3]), P(gs), P(ge), P(gP))
    assert _rel(gs, gs_r.numpy()) < 1e-10
    assert _rel(ge, ge_r.numpy()) < 1e-10
    assert _rel(gP, gP_r.numpy()) < 1e-10
    # d / d (n_in, n_out): "value" mode reads the indices as tensors (operation.py:268-272).  Only
    # rays with a child see them; mirrors (n_in = 0) and totally reflected rays give exactly zero
    ni, no = torch.tensor(n_in, requires_grad=True), torch.tensor(n_out, requires_grad=True)
    o = geom.snells_law_3D(st[:, 0], st[:, 1], st[:, 2], x, y, z, norm, ni, no, L)
    ce = torch.stack(o[3:], 1)
    gni, gno = torch.autograd.grad((cm * ce * torch.tensor(g[3])).sum(), [ni, no])
    gn = np.zeros((n, 2))
    host_math.hm_adjoint3d_n(I(n), P(s), P(e), P(P9), P(ruu), P(child), P(n_in), P(n_out), D(L),
                             P(g[0]), P(gh), P(g[3]), P(gs), P(ge), P(gP), P(gn))
    assert _rel(gn[:, 0], gni.numpy()) < 1e-10 and _rel(gn[:, 1], gno.numpy()) < 1e-10
    assert np.abs(gni.numpy()).max() > 0 and (gn[n_in == 0.0] == 0).all()


def test_segment_exact_and_adjoint2d(host_math):
    rng = np.random.default_rng(2)
    n = 2000
    L = 1.3
    seg = rng.normal(size=(n, 4)) * 2
    tpar = rng.uniform(0.05, 0.95, size=(n, 1))
    hit = seg[:, :2] + tpar * (seg[:, 2:] - seg[:, :2])
    s = hit + rng.normal(size=(n, 2)) * 2
    e = s + (hit - s) * rng.uniform(0.3, 2.5, size=(n, 1))
    ru, su, xy, va = np.zeros(n), np.zeros(n), np.zeros((n, 2)), np.zeros(n, np.uint8)
    host_math.hm_exact_segment(I(n), P(s), P(e), P(seg), D(1e-10), D(1e-10), D(1e-10), P(ru),
                               P(su), P(xy), P(va))
    x, y, valid, u, v = geom.raw_line_intersect(s[:, 0], s[:, 1], e[:, 0], e[:, 1], seg[:, 0],
                                                seg[:, 1], seg[:, 2], seg[:, 3], 1e-10)
    assert np.array_equal(ru, u.numpy()) and np.array_equal(su, v.numpy()) and va.all()
    n_in = rng.choice([1.0, 1.5, 0.0, 1.33], size=n)
    n_out = rng.choice([1.0, 1.5, 1.2], size=n)
    child = rng.integers(0, 2, size=n).astype(np.uint8)
    st, et, pt = [torch.tensor(a, requires_grad=True) for a in (s, e, seg)]
    x, y, valid, u, v = geom.raw_line_intersect(st[:, 0], st[:, 1], et[:, 0], et[:, 1], pt[:, 0],
                                                pt[:, 1], pt[:, 2], pt[:, 3], 1e-10)
    norm = torch.atan2(pt[:, 3] - pt[:, 1], pt[:, 2] - pt[:, 0]) + PI / 2
    # (the Snell step is applied to the rows that have a child only: in the reference's tape a row
    # that goes through snells_law_2D gets asin's 0 * NaN under total internal reflection even
    # when nothing downstream depends on it)
    ci = torch.tensor(np.nonzero(child)[0])
    g = [rng.normal(size=(n, 2)) for _ in range(4)]
    hT = torch.stack([x, y], 1)

    def total(finite):
        o = geom.snells_law_2D(st[ci, 0], st[ci, 1], x[ci], y[ci], norm[ci], torch.tensor(n_in)[ci],
                               torch.tensor(n_out)[ci], L, finite_tir_gradient=finite)
        g2, g3 = torch.tensor(g[2])[ci], torch.tensor(g[3])[ci]
        return (st * torch.tensor(g[0])).sum() + (hT * torch.tensor(g[1])).sum() + (
            o[0] * g2[:, 0] + o[1] * g2[:, 1] + o[2] * g3[:, 0] + o[3] * g3[:, 1]).sum()
    gr = torch.autograd.grad(total(False), [st, et, pt], retain_graph=True)
    gs, ge, gp = np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 5))
    gh = g[1] + child[:, None] * g[2]
    uu = u.detach().numpy().copy()
    host_math.hm_adjoint2d(I(n), P(s), P(e), P(seg), ctypes.c_int(4), ctypes.c_int(0), P(uu),
                           P(child), P(n_in), P(n_out), D(L), P(g[0]), P(gh), P(g[3]), P(gs),
                           P(ge), P(gp))
    # totally reflected rays: NaN on both sides (geometry.py:640-646: asin in the unselected branch)
    bad = ~np.isfinite(gr[0].numpy()).all(axis=1)
    assert 0 < bad.sum() < n // 4
    for got, want in ((gs, gr[0].numpy()), (ge, gr[1].numpy()), (gp[:, :4], gr[2].numpy())):
        assert np.array_equal(~np.isfinite(got).all(axis=1), bad)
        assert np.isnan(got[bad]).all() and np.isnan(want[bad]).all()
        assert _rel(got[~bad], want[~bad]) < 1e-10
    # ... and the opt-in finite form (tfrt_scene2d.finite_tir_gradient) against the oracle's
    gr = torch.autograd.grad(total(True), [st, et, pt])
    host_math.hm_adjoint2d_finite(I(n), P(s), P(e), P(seg), ctypes.c_int(4), ctypes.c_int(0), P(uu),
                                  P(child), P(n_in), P(n_out), D(L), P(g[0]), P(gh), P(g[3]), P(gs),
                                  P(ge), P(gp))
    assert _rel(gs, gr[0].numpy()) < 1e-10 and _rel(ge, gr[1].numpy()) < 1e-10
    assert _rel(gp[:, :4], gr[2].numpy()) < 1e-10


def test_arc_exact_matches_oracle(host_math):
    rng = np.random.default_rng(3)
    n = 600
    arc = np.stack([rng.normal(size=n) * 3, rng.normal(size=n) * 3, rng.uniform(-PI, PI, n),
                    rng.uniform(-PI, PI, n), rng.uniform(0.3, 2, n) * rng.choice([-1, 1], n)], 1)
    ang = rng.uniform(-PI, PI, n)
    hit = arc[:, :2] + np.abs(arc[:, 4:5]) * np.stack([np.cos(ang), np.sin(ang)], 1)
    s = hit + rng.normal(size=(n, 2)) * 2
    e = s + (hit - s) * rng.uniform(0.3, 2.5, size=(n, 1))
    ru, au, xy = np.zeros(n), np.zeros(n), np.zeros((n, 2))
    va, nm = np.zeros(n, np.uint8), np.zeros(n)
    host_math.hm_exact_arc(I(n), P(s), P(e), P(arc), D(1e-10), D(1e-10), P(ru), P(au), P(xy), P(va), P(nm))
    seen = 0
    for i in range(n):
        tt = lambda a: torch.tensor(a[i:i + 1])
        x, y, valid, ray_u, arc_u, _, g = tracer.arc_intersection(
            tt(s[:, 0]), tt(s[:, 1]), tt(e[:, 0]), tt(e[:, 1]), tt(arc[:, 0]), tt(arc[:, 1]),
            tt(arc[:, 2]), tt(arc[:, 3]), tt(arc[:, 4]), 1e-10, 1e-10, 1e-10)
        assert bool(valid[0]) == bool(va[i])
        if va[i]:
            seen += 1
            assert abs(float(ray_u[0]) - ru[i]) < 1e-13 and abs(float(arc_u[0]) - au[i]) < 1e-13
            assert abs(float(tracer.get_arc_norm(tt(arc[:, 4]), arc_u, g)[0]) - nm[i]) < 1e-13
    assert 100 < seen < n


def test_arc_hit_variant_equals_exact_arc_on_every_valid_hit(host_math):
    """csrc/trace_math2d.h::exact_arc_hit (one atan2 where exact_arc takes three) against exact_arc
    itself: the same validity everywhere, the same bits on every valid hit -- rays from outside
    and inside the circles, tangents, short arcs the first root misses, negative radii."""
    rng = np.random.default_rng(31)
    n = 40_000
    arc = np.stack([rng.normal(size=n) * 3, rng.normal(size=n) * 3, rng.uniform(-PI, PI, n),
                    rng.uniform(-PI, PI, n), rng.uniform(0.3, 2, n) * rng.choice([-1, 1], n)], 1)
    arc[:4000, 3] = arc[:4000, 2] + rng.uniform(0.05, 0.6, 4000)      # short arcs
    ang = rng.uniform(-PI, PI, n)
    hit = arc[:, :2] + np.abs(arc[:, 4:5]) * np.stack([np.cos(ang), np.sin(ang)], 1)
    s = hit + rng.normal(size=(n, 2)) * 2
    inside = rng.random(n) < 0.4                                         # start inside the circle
    s[inside] = arc[inside, :2] + rng.uniform(-0.5, 0.5, (int(inside.sum()), 2)) * np.abs(arc[inside, 4:5])
    e = s + (hit - s) * rng.uniform(0.3, 2.5, size=(n, 1))
    tang = slice(n - 2000, n)                                            # tangent lines
    tdir = np.stack([-np.sin(ang[tang]), np.cos(ang[tang])], 1)
    s[tang] = hit[tang] - tdir * 1.5
    e[tang] = hit[tang] + tdir * 0.5
    out = []
    for fn, extra in ((host_math.hm_exact_arc, True), (host_math.hm_exact_arc_hit, False)):
        ru, au, xy = np.zeros(n), np.zeros(n), np.zeros((n, 2))
        va, nm = np.zeros(n, np.uint8), np.zeros(n)
        args = [I(n), P(s), P(e), P(arc), D(1e-10), D(1e-10), P(ru), P(au), P(xy), P(va)]
        fn(*(args + ([P(nm)] if extra else [])))
        out.append((ru, au, xy, va))
    (ru0, au0, xy0, va0), (ru1, au1, xy1, va1) = out
    assert np.array_equal(va0, va1)
    v = va0.astype(bool)
    assert 0.2 * n < v.sum() < 0.95 * n
    assert np.array_equal(ru0[v], ru1[v]) and np.array_equal(au0[v], au1[v])
    assert np.array_equal(xy0[v], xy1[v])


def test_snell3d_restatement_is_bit_identical_to_the_oracle(host_math):
    """csrc/trace_math.h::snell3d + advance against oracle.geom.snells_law_3D (geometry.py:715-753)
    on 20,000 random rays incl. TIR, mirrors and n_out = 0: every bit equal.  (The same comparison
    runs on the device in tests/test_gpu_stress.py.)"""
    rng = np.random.default_rng(77)
    n = 20_000
    s = rng.uniform(-3, 3, (n, 3)) * 10 ** rng.uniform(-2, 2, (n, 1))
    h = s + rng.standard_normal((n, 3)) * 10 ** rng.uniform(-2, 1, (n, 1))
    norm = rng.standard_normal((n, 3)) * 10 ** rng.uniform(-3, 3, (n, 1))
    n_in = rng.uniform(1.0, 1.8, n)
    n_out = rng.uniform(1.0, 1.8, n)
    n_in[:500] = 0.0
    n_out[500:800] = 0.0
    L = 0.37
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    want = torch.stack(geom.snells_law_3D(
        t(s[:, 0]), t(s[:, 1]), t(s[:, 2]), t(h[:, 0]), t(h[:, 1]), t(h[:, 2]), t(norm), t(n_in),
        t(n_out), L)).numpy()
    got = np.zeros((6, n))
    host_math.hm_snell3d_norm(I(n), P(s), P(h), P(norm), P(n_in), P(n_out), D(L), P(got))
    assert int((got != want).any(axis=0).sum()) == 0


def test_oracle_square_root_is_correctly_rounded():
    """oracle.geom.sqrt = IEEE sqrt (what TensorFlow's CPU kernels compute); torch.sqrt of this
    image is not, which is why the oracle does not use it."""
    rng = np.random.default_rng(5)
    x = rng.uniform(1.0, 2.0, 200_000) * 2.0 ** rng.integers(-40, 40, 200_000)
    xt = torch.tensor(x, requires_grad=True)
    y = geom.sqrt(xt)
    assert np.array_equal(y.detach().numpy(), np.sqrt(x))
    # correctly rounded: x lies between the squares of the midpoints to y's two neighbours
    yn = y.detach().numpy()
    lo, hi = np.nextafter(yn, 0.0), np.nextafter(yn, np.inf)
    import decimal
    decimal.getcontext().prec = 80
    d = lambda v: decimal.Decimal(float(v))
    for k in rng.choice(x.size, 300, replace=False):
        assert ((d(lo[k]) + d(yn[k])) / 2) ** 2 <= d(x[k]) <= ((d(hi[k]) + d(yn[k])) / 2) ** 2
    (g,) = torch.autograd.grad(y.sum(), [xt])
    np.testing.assert_allclose(g.numpy(), 0.5 / np.sqrt(x), rtol=1e-15)


def test_face_normal_restatement_is_bit_identical_to_the_oracle(host_math):
    """csrc/trace_math.h::face_normal against oracle.tracer.faces_from_vertices
    (boundaries.py:918-923: cross, then x / sqrt(sum x^2)), 50,000 random triangles over twelve
    orders of magnitude: every bit equal."""
    rng = np.random.default_rng(0)
    n = 50_000
    P9 = rng.standard_normal((n, 9)) * 10 ** rng.uniform(-3, 3, (n, 1))
    want = tracer.faces_from_vertices(torch.tensor(P9).reshape(-1, 3),
                                      np.arange(3 * n).reshape(n, 3))["norm"].numpy()
    got = np.zeros((n, 3))
    host_math.hm_face_normal(I(n), P(P9), P(got))
    assert int((got != want).any(axis=1).sum()) == 0
