"""
Adversarial scenes for the discrete decisions of the 3-D path: random triangle soups with nasty
scales and offsets, grazing rays, and a third of the faces in ONE plane, so that overlapping
coplanar faces tie in ``ray_u`` up to the last bit.  Which of two such faces wins (tf.argmin's
first-index rule, engine.py:1149) then depends on the last bit of the ray that enters the pass,
i.e. on every operation of the previous pass's Snell step being rounded like the reference's
eager float64 ops (geometry.py:715-753).

History: seed 35 of this generator (ray 453) used to end on face 32 on the device and on face 57
in the oracle.  Cause: the child ray's end ``hit + L * w`` (geometry.py:751-752) was written
inline in k_react3d and contracted to one fma by hipcc's default -ffp-contract=fast, so the child
differed from the reference's in the last bit of its end point; it now goes through
trace_math.h::advance (product and sum rounded separately).  What is guaranteed and asserted
here: with float64 ray state the children of a pass are BIT-IDENTICAL to the oracle's, and
therefore classes, order and hit faces of multi-pass traces are identical too, ties included.
"""
import numpy as np
import pytest
import torch

from oracle import geom, tracer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


def _soup(seed):
    """The generator of scratch/stress_oracle.py (kept verbatim in its random draws so that the
    seeds keep their meaning)."""
    rng = np.random.default_rng(5000 + seed)
    n_faces = int(rng.choice([64, 97, 300, 640]))
    n_rays = int(rng.choice([50, 700, 2500]))
    scale = 10 ** rng.uniform(-3, 3)
    offset = rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(0, 2.5) * (rng.random() < 0.5)
    centre = rng.uniform(-1, 1, (n_faces, 1, 3))
    size = 10 ** rng.uniform(-2.5, -0.2, (n_faces, 1, 1))
    tri = (centre + size * rng.standard_normal((n_faces, 3, 3))) * scale + offset
    coplanar = bool(rng.random() < 0.5)
    if coplanar:
        tri[: n_faces // 3, :, 2] = offset[2] + 0.1 * scale
    P = torch.tensor(tri.reshape(n_faces, 9), dtype=torch.float64)
    cat = torch.zeros(n_faces, dtype=torch.int64)
    cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1
    cat[int(0.9 * n_faces):] = 2
    n_in = torch.tensor(rng.uniform(1.0, 1.7, n_faces))
    n_out = torch.tensor(rng.uniform(1.0, 1.7, n_faces))
    s = rng.uniform(-1.5, 1.5, (3, n_rays)) * scale + offset[:, None]
    d = rng.standard_normal((3, n_rays))
    if rng.random() < 0.5:
        d[2] *= 1e-3
    e = s + d * scale * 10 ** rng.uniform(-2, 0.5)
    rays = torch.tensor(np.concatenate([s, e]), dtype=torch.float64)
    return dict(P=P, cat=cat, n_in=n_in, n_out=n_out, rays=rays, L=float(scale),
                coplanar=coplanar)


def _oracle_system(sc):
    def sub(mask):
        verts = sc["P"][mask].reshape(-1, 3)
        d = tracer.faces_from_vertices(verts, torch.arange(verts.shape[0]).reshape(-1, 3))
        d["n_in"], d["n_out"] = sc["n_in"][mask], sc["n_out"][mask]
        d["face_index"] = torch.nonzero(mask).reshape(-1).double()
        return d
    cat = sc["cat"]
    return tracer.System(3, optical=sub(cat == 0), stop=sub(cat == 1), target=sub(cat == 2))


def _oracle_trace(sc, passes):
    src = {n: sc["rays"][i] for i, n in enumerate(NAMES)}
    src["ray_id"] = torch.arange(sc["rays"].shape[1], dtype=torch.float64)
    return tracer.ray_trace(_oracle_system(sc), src, max_iterations=passes, inherit=("ray_id",),
                            index_type="value", new_ray_length=sc["L"],
                            flags=dict(compile_dead_rays=True, compile_stopped_rays=True))


def _gpu_trace(sc, passes, clustered=True):
    from tensorflowraytrace_amd import ops, _lib
    fv = sc["P"].to(DEV)
    args = ops.Scene3DArgs(fv, sc["cat"].int().to(DEV), n_in=sc["n_in"].to(DEV),
                           n_out=sc["n_out"].to(DEV),
                           cluster_order=ops.cluster_order(fv) if clustered else None)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    return ops.trace3d(sc["rays"].to(DEV), fv, args, max_passes=passes, flags=flags,
                       new_ray_length=sc["L"])


def _block(rayset):
    return torch.stack([rayset[n] for n in NAMES])


def test_snell3d_seam_is_bit_identical_to_the_oracle():
    """geometry.py:715-753 on 20,000 random rays / normals / index pairs incl. total internal
    reflection, mirrors (n_in = 0) and n_out = 0: every output bit equal."""
    from tensorflowraytrace_amd import ops
    rng = np.random.default_rng(77)
    n = 20_000
    s = rng.uniform(-3, 3, (n, 3)) * 10 ** rng.uniform(-2, 2, (n, 1))
    h = s + rng.standard_normal((n, 3)) * 10 ** rng.uniform(-2, 1, (n, 1))
    norm = rng.standard_normal((n, 3)) * 10 ** rng.uniform(-3, 3, (n, 1))   # not unit length
    n_in = rng.uniform(1.0, 1.8, n)
    n_out = rng.uniform(1.0, 1.8, n)
    n_in[:500] = 0.0
    n_out[500:800] = 0.0
    L = 0.37
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    ref = geom.snells_law_3D(t(s[:, 0]), t(s[:, 1]), t(s[:, 2]), t(h[:, 0]), t(h[:, 1]), t(h[:, 2]),
                             t(norm), t(n_in), t(n_out), L)
    d = lambda a: torch.tensor(a, dtype=torch.float64, device=DEV)
    out = ops.snell3d(d(s[:, 0]), d(s[:, 1]), d(s[:, 2]), d(h[:, 0]), d(h[:, 1]), d(h[:, 2]),
                      d(norm), d(n_in), d(n_out), L).cpu()
    want = torch.stack(ref)
    assert bool(torch.isfinite(want).all())
    differing = int((out != want).any(dim=0).sum())
    assert differing == 0, f"{differing} of {n} rays differ in some bit"


@pytest.mark.parametrize("seed", [35, 3, 8, 21])
def test_children_of_one_pass_are_bit_identical(seed):
    """One pass (intersect -> project -> react) in float64 state: hit faces, classes and the
    child rays handed to the next pass equal the oracle's bit for bit."""
    sc = _soup(seed)
    out = _gpu_trace(sc, 1)
    ref = _oracle_trace(sc, 1)
    child = ref["unfinished"]
    assert child and child["x_start"].shape[0] > 0
    assert torch.equal(out["unfinished_id"].cpu().long(), child["ray_id"].long())
    assert torch.equal(out["unfinished"].cpu(), _block(child))
    act = ref["active"]
    assert torch.equal(out["active"].cpu(), _block(act))          # projected ends = hit points


@pytest.mark.parametrize("seed", [35, 0, 5, 11, 17, 28, 33, 39, 44])
def test_soup_traces_equal_the_oracle_including_coplanar_ties(seed):
    """Three passes: every class holds the same rays in the same order with the same geometry,
    in the hierarchy mode and in the all-pairs mode (seed 35: the former mismatch)."""
    sc = _soup(seed)
    ref = _oracle_trace(sc, 3)
    for clustered in (True, False):
        out = _gpu_trace(sc, 3, clustered)
        for cls in ("finished", "active", "stopped", "dead"):
            r = ref[cls]
            n_ref = r["x_start"].shape[0] if r else 0
            assert out[cls].shape[1] == n_ref, (cls, clustered)
            if not n_ref:
                continue
            assert torch.equal(out[cls + "_id"].cpu().long(), r["ray_id"].long()), (cls, clustered)
            assert torch.equal(out[cls].cpu(), _block(r)), (cls, clustered)
    if seed == 35:
        assert sc["coplanar"]
        dead_ids = out["dead_id"].cpu().long()
        row = int(torch.nonzero(dead_ids == 453)[0])
        assert torch.equal(out["dead"].cpu()[:, row], _block(ref["dead"])[:, row])


@pytest.mark.parametrize("seed", [16, 35, 2])
def test_soup_traces_with_a_large_size_epsilion_equal_the_oracle(seed):
    """size_epsilion = 0.3: trig_u, trig_v >= -0.3, trig_u + trig_v <= 1.3 accepts hits up to
    0.3 |2 E1 - E2| outside the triangle -- more than the 0.3 (|E1| + |E2|) the bounding spheres
    were inflated by until round 3 (one ray of soup 16 lost its hit to the filter).  Every trace
    mode against the oracle's exact all-pairs result."""
    from tensorflowraytrace_amd import ops, _lib
    sc = _soup(seed)
    eps = (1e-6, 0.3, 1e-10)
    system = _oracle_system(sc)
    system.eps = eps
    src = {n: sc["rays"][i] for i, n in enumerate(NAMES)}
    src["ray_id"] = torch.arange(sc["rays"].shape[1], dtype=torch.float64)
    ref = tracer.ray_trace(system, src, max_iterations=3, inherit=("ray_id",), index_type="value",
                           new_ray_length=sc["L"],
                           flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    fv = sc["P"].to(DEV)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    for mode in ("all-pairs", "hierarchy", "coherent"):
        args = ops.Scene3DArgs(fv, sc["cat"].int().to(DEV), n_in=sc["n_in"].to(DEV),
                               n_out=sc["n_out"].to(DEV),
                               cluster_order=None if mode == "all-pairs" else ops.cluster_order(fv),
                               coherent_rays=mode == "coherent")
        args.eps = eps
        out = ops.trace3d(sc["rays"].to(DEV), fv, args, max_passes=3, flags=flags,
                          new_ray_length=sc["L"])
        for cls in ("finished", "active", "stopped", "dead"):
            r = ref[cls]
            n_ref = r["x_start"].shape[0] if r else 0
            assert out[cls].shape[1] == n_ref, (cls, mode)
            if n_ref:
                assert torch.equal(out[cls + "_id"].cpu().long(), r["ray_id"].long()), (cls, mode)
                assert torch.equal(out[cls].cpu(), _block(r)), (cls, mode)


@pytest.mark.parametrize("op,name", [(0, "div"), (1, "sqrt"), (2, "1/sqrt"), (3, "mul+add")])
def test_device_float64_primitives_are_correctly_rounded(op, name):
    """tfrt_selftest_f64 against numpy (IEEE division / sqrt on the host), 2M random operands over
    many binades: every result bit equal."""
    import ctypes
    from tensorflowraytrace_amd import _lib, ops
    rng = np.random.default_rng(op)
    n = 2_000_000
    a = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-40, 40, n)
    b = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-40, 40, n)
    if op == 0:
        want = a / b
    elif op == 1:
        want = np.sqrt(a)
    elif op == 2:
        want = 1.0 / np.sqrt(a)
    else:
        want = a + b * b
    ta, tb = torch.tensor(a, device=DEV), torch.tensor(b, device=DEV)
    out = torch.empty_like(ta)
    _lib.check(_lib.lib().tfrt_selftest_f64(op, n, ops._p(ta), ops._p(tb), ops._p(out),
                                            ops._stream(ta)), "tfrt_selftest_f64")
    got = out.cpu().numpy()
    bad = np.nonzero(got != want)[0]
    ulp = np.abs(got[bad].view(np.int64) - want[bad].view(np.int64)).max() if bad.size else 0
    assert bad.size == 0, f"{name}: {bad.size} of {n} results differ (max {ulp} ulp)"


@pytest.mark.parametrize("op,name", [(4, "adj_rcp"), (5, "adj_rsqrt")])
def test_reverse_sweep_reciprocals_against_ieee(op, name):
    """The DEVICE branch of trace_math.h's adj_rcp / adj_rsqrt (hardware estimate + two Newton steps;
    tests/host_math only sees the host branch, which is the plain IEEE operation): against numpy on
    1M operands over 160 binades, denormal results and operands included.  They are not correctly
    rounded by design (the reverse sweep's tolerance is 1e-8): the bound is 2 ulp, and the special
    values must come out as IEEE gives them wherever the sweep can meet them."""
    from tensorflowraytrace_amd import _lib, ops
    rng = np.random.default_rng(40 + op)
    n = 1_000_000
    a = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-80, 80, n)
    if op == 4:
        a *= rng.choice([-1.0, 1.0], n)
    # the ends of the range: large operands (results denormal) and tiny ones
    a[:1000] = rng.uniform(1.0, 2.0, 1000) * 2.0 ** rng.integers(990, 1023, 1000)
    a[1000:2000] = rng.uniform(1.0, 2.0, 1000) * 2.0 ** rng.integers(-1022, -990, 1000)
    want = 1.0 / a if op == 4 else 1.0 / np.sqrt(a)
    ta = torch.tensor(a, device=DEV)
    out = torch.empty_like(ta)
    _lib.check(_lib.lib().tfrt_selftest_f64(op, n, ops._p(ta), None, ops._p(out), ops._stream(ta)),
               "tfrt_selftest_f64")
    got = out.cpu().numpy()
    normal = np.abs(want) >= 2.0 ** -1022
    ulp = np.abs(got[normal].view(np.int64) - want[normal].view(np.int64))
    assert ulp.max() <= 2, f"{name}: {int((ulp > 2).sum())} results off by more than 2 ulp (max {ulp.max()})"
    assert (ulp == 0).mean() > 0.5           # (most results are the correctly rounded ones)
    # denormal results: absolute error below two denormal steps
    assert np.abs(got[~normal] - want[~normal]).max(initial=0.0) <= 2 * 4.9406564584124654e-324 * 2 ** 1
    # special operands
    sp = torch.tensor([np.inf, 4.0, 1.0], dtype=torch.float64, device=DEV)
    o2 = torch.empty_like(sp)
    _lib.check(_lib.lib().tfrt_selftest_f64(op, 3, ops._p(sp), None, ops._p(o2), ops._stream(sp)),
               "tfrt_selftest_f64")
    assert o2.cpu().tolist()[1:] == ([0.25, 1.0] if op == 4 else [0.5, 1.0])
