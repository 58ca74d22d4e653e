"""
tests/golden/reference_geometry.npz holds the outputs of the reference's OWN tfrt/geometry.py
source (executed in the build container under the minimal TensorFlow stand-in of tests/tf_shim,
see tests/golden/make_reference_golden.py) on seeded random inputs: raw_line_intersect,
raw_line_triangle_intersect, raw_line_circle_intersect, angle_in_interval, snells_law_2D,
snells_law_3D (tfrt/geometry.py:96-802).

* the oracle restatement reproduces every output BIT FOR BIT (same formulas, same order, same
  masking, same arithmetic): a transcription error in oracle/geom.py would show here;
* on the GPU the HIP entry points reproduce the algebraic outputs (line x line, line x triangle,
  Snell 3-D) bit for bit and the outputs that go through atan2 / sin / cos / asin (device libm vs
  host libm) to 1e-12; valid masks identical.
"""
import os

import numpy as np
import pytest
import torch

from oracle import geom

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_geometry.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def _t(a, dev="cpu"):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)


def test_oracle_reproduces_the_reference_source_bit_for_bit(g):
    eps = 1e-10
    x, y, valid, u, v = geom.raw_line_intersect(*[_t(r) for r in g["li"]], eps)
    assert np.array_equal(valid.numpy(), g["li_valid"]) and 0 < (~g["li_valid"]).sum() < 300
    assert np.array_equal(torch.stack([x, y, u, v]).numpy(), g["li_out"])
    res = geom.raw_line_triangle_intersect(*[_t(r) for r in g["tri_rays"]], *[_t(r) for r in g["tri"]], eps)
    assert np.array_equal(res[3].numpy(), g["tri_valid"])
    assert np.array_equal(torch.stack([res[0], res[1], res[2], res[4], res[5], res[6]]).numpy(), g["tri_out"])
    plus, minus = geom.raw_line_circle_intersect(*[_t(r) for r in g["circ_lines"]],
                                                 *[_t(r) for r in g["circ"]], eps)
    for name, root in (("plus", plus), ("minus", minus)):
        assert np.array_equal(root["valid"].numpy(), g[f"circ_{name}_valid"]), name
        got = torch.stack([root["x"], root["y"], root["u"], root["v"]]).numpy()
        assert np.array_equal(got, g[f"circ_{name}"]), name
    assert 0 < g["circ_plus_valid"].sum() < g["circ_plus_valid"].size       # hits and misses
    assert np.array_equal(geom.angle_in_interval(*[_t(r) for r in g["ang"]]).numpy(), g["ang_out"])
    L = float(g["new_ray_length"])
    o2 = geom.snells_law_2D(*[_t(r) for r in g["sn2_rays"]], _t(g["sn2_norm"]), _t(g["sn_n"][0]),
                            _t(g["sn_n"][1]), L)
    assert np.array_equal(torch.stack(list(o2)).numpy(), g["sn2_out"])
    o3 = geom.snells_law_3D(*[_t(r) for r in g["sn3_rays"]], _t(g["sn3_norm"]), _t(g["sn_n"][0]),
                            _t(g["sn_n"][1]), L)
    assert np.array_equal(torch.stack(list(o3)).numpy(), g["sn3_out"])


@pytest.mark.gpu
def test_hip_geometry_reproduces_the_reference_source(g):
    from tensorflowraytrace_amd import ops
    dev = "cuda:0"
    t = lambda a: _t(a, dev)
    eps = 1e-10
    x, y, valid, u, v = ops.line_intersect([t(r) for r in g["li"][:4]], [t(r) for r in g["li"][4:]],
                                           eps, grid=False)
    assert np.array_equal(valid.cpu().numpy(), g["li_valid"])
    assert np.array_equal(torch.stack([x, y, u, v]).cpu().numpy(), g["li_out"])
    x, y, z, valid, ru, tu, tv = ops.line_triangle_intersect(
        [t(r) for r in g["tri_rays"]], [t(r) for r in g["tri"]], eps, grid=False)
    assert np.array_equal(valid.cpu().numpy(), g["tri_valid"])
    assert np.array_equal(torch.stack([x, y, z, ru, tu, tv]).cpu().numpy(), g["tri_out"])
    plus, minus = ops.line_circle_intersect([t(r) for r in g["circ_lines"]], [t(r) for r in g["circ"]],
                                            eps, grid=False)
    for name, root in (("plus", plus), ("minus", minus)):
        assert np.array_equal(root["valid"].cpu().numpy(), g[f"circ_{name}_valid"]), name
        got = torch.stack([root["x"], root["y"], root["u"], root["v"]]).cpu().numpy()
        np.testing.assert_allclose(got, g[f"circ_{name}"], rtol=0, atol=1e-12, err_msg=name)
    L = float(g["new_ray_length"])
    s3 = g["sn3_rays"]
    out = ops.snell3d(*[t(s3[i]) for i in range(6)], t(g["sn3_norm"]), t(g["sn_n"][0]),
                      t(g["sn_n"][1]), L).cpu().numpy()
    assert np.array_equal(out, g["sn3_out"])
    s2 = g["sn2_rays"]
    out = ops.snell2d(*[t(s2[i]) for i in range(4)], t(g["sn2_norm"]), t(g["sn_n"][0]),
                      t(g["sn_n"][1]), L).cpu().numpy()
    np.testing.assert_allclose(out, g["sn2_out"], rtol=0, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# tests/golden/reference_trace3d.npz: a 4-pass 3-D trace of the lens scene (1,200 rays x 246 faces)
# and its parameter gradients, produced by the reference's own engine / operation / materials /
# geometry modules executed under tests/tf_shim (tests/golden/make_reference_trace_golden.py)

TRACE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_trace3d.npz")
NAMES = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


def test_oracle_trace_reproduces_the_reference_engine_bit_for_bit():
    import oracle_util
    from oracle import tracer
    g = np.load(TRACE)
    scene = {k: g[k] for k in g.files}
    system, (p_f, p_b), _ = oracle_util.lens_oracle(scene)
    ref = tracer.ray_trace(system, oracle_util.source_dict(g["rays"], g["wavelength"]),
                           max_iterations=int(g["passes"]), inherit=("wavelength", "ray_id"),
                           flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    for cls in ("finished", "active"):
        assert np.array_equal(ref[cls]["ray_id"].numpy().astype(np.int64), g[cls + "_id"]), cls
        assert np.array_equal(oracle_util.block(ref[cls]), g[cls]), cls      # every bit
    assert g["dead"].shape[1] == 0 and (not ref["dead"] or ref["dead"]["x_start"].shape[0] == 0)
    fin = ref["finished"]
    goal = torch.tensor(g["goal"])[fin["ray_id"].long()]
    loss = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
    assert float(loss.detach()) == float(g["loss"])
    g_f, g_b = torch.autograd.grad(loss, [p_f, p_b])
    for got, want in ((g_f.numpy(), g["grad_front"]), (g_b.numpy(), g["grad_back"])):
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 0.0), (torch.float32, 1e-5)])
def test_hip_trace_reproduces_the_reference_engine(dtype, tol):
    """float64 ray state: classes, order and every coordinate of the finished and active sets equal
    the reference engine's output bit for bit; gradients to 1e-8.  float32 state: 1e-5."""
    from tensorflowraytrace_amd import ops, _lib
    from test_gpu_trace3d import _gpu_scene
    g = np.load(TRACE)
    scene = {k: g[k] for k in g.files}
    for cluster in ("group", False):
        src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype, cluster=cluster)
        flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
        out = ops.trace3d(src, fv, sc, max_passes=int(g["passes"]), flags=flags)
        assert out["dead"].shape[1] == 0 and out["stopped"].shape[1] == 0
        for cls in ("finished", "active"):
            assert np.array_equal(out[cls + "_id"].cpu().numpy().astype(np.int64), g[cls + "_id"]), cls
            got = out[cls].detach().cpu().double().numpy()
            if tol == 0.0:
                assert np.array_equal(got, g[cls]), (cls, cluster)
            else:
                assert np.abs(got - g[cls]).max() / max(1.0, np.abs(g[cls]).max()) <= tol
        fin = out["finished"]
        goal = torch.tensor(g["goal"], device=fin.device)[out["finished_id"].long()]
        loss = ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
        g_f, g_b = torch.autograd.grad(loss, [p_f, p_b])
        gtol = 1e-8 if tol == 0.0 else tol
        assert abs(float(loss.detach()) - float(g["loss"])) <= max(gtol, 1e-12) * float(g["loss"])
        for got, want in ((g_f.cpu().numpy(), g["grad_front"]), (g_b.cpu().numpy(), g["grad_back"])):
            assert np.abs(got - want).max() <= gtol * np.abs(want).max()


# ---------------------------------------------------------------------------------------------
# tests/golden/reference_trace2d.npz: 4-pass 2-D traces (arcs only / segments only / both) by the
# reference's own OpticalSystem2D + OpticalEngine (tests/golden/make_reference_trace2d_golden.py)

TRACE2D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_trace2d.npz")
GEO2 = ("x_start", "y_start", "x_end", "y_end")


def _sets_2d(g, tag):
    sets = {}
    for key in g.files:
        if key.startswith(tag + "__"):
            _, name, field = key.split("__")
            v = g[key]
            sets.setdefault(name, {})[field] = torch.tensor(v)
    return sets


def _oracle_2d(g, tag, bug_compatible=False):
    from oracle import tracer
    sets = _sets_2d(g, tag)
    osys = tracer.System(2, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"],
                                       tracer.MATERIALS["reflective"]], **sets)
    rays, wl = g[tag + "_rays"], g[tag + "_wl"]
    src = {n: torch.tensor(rays[i]) for i, n in enumerate(GEO2)}
    src["wavelength"] = torch.tensor(wl)
    src["ray_id"] = torch.arange(rays.shape[1], dtype=torch.float64)
    return tracer.ray_trace(osys, src, max_iterations=4, inherit=("wavelength", "ray_id"),
                            flags=dict(compile_dead_rays=True, compile_stopped_rays=True),
                            bug_compatible=bug_compatible), sets


@pytest.mark.parametrize("tag", ["arc", "seg", "both"])
def test_oracle_2d_trace_reproduces_the_reference_engine_bit_for_bit(tag):
    """Unmixed scenes: plain oracle.  The mixed scene: the reference pairs the reacting rays
    [segment hits, arc hits] with boundary data ordered [arc, segment] (engine.py:1958-1965); the
    oracle reproduces that output exactly with ``bug_compatible=True``."""
    import oracle_util
    g = np.load(TRACE2D)
    ref, _ = _oracle_2d(g, tag, bug_compatible=(tag == "both"))
    for cls in ("finished", "active", "stopped", "dead"):
        want = g[f"{tag}_{cls}"]
        got = oracle_util.block(ref[cls], dim=2) if ref[cls] else np.zeros((4, 0))
        assert got.shape == want.shape, (tag, cls)
        if want.size:
            assert np.array_equal(ref[cls]["ray_id"].numpy().astype(np.int64), g[f"{tag}_{cls}_id"])
            assert np.array_equal(got, want), (tag, cls)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["arc", "seg"])
def test_hip_2d_trace_reproduces_the_reference_engine(tag):
    """float64 ray state: same rays in the same classes and order; coordinates to 1e-9 (atan2 /
    sin / cos / asin come from the device's libm)."""
    from tensorflowraytrace_amd import ops, _lib
    import test_gpu_trace2d as t2
    g = np.load(TRACE2D)
    sets = _sets_2d(g, tag)
    scene, _, _ = t2._gpu_scene(sets, g[tag + "_wl"])
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src = torch.tensor(g[tag + "_rays"], dtype=torch.float64, device="cuda:0")
    out = ops.trace2d(src, scene, max_passes=4, flags=flags)
    for cls in ("finished", "active", "stopped", "dead"):
        want = g[f"{tag}_{cls}"]
        assert out[cls].shape[1] == want.shape[1], (tag, cls)
        if want.size:
            assert np.array_equal(out[cls + "_id"].cpu().numpy().astype(np.int64), g[f"{tag}_{cls}_id"])
            np.testing.assert_allclose(out[cls].detach().cpu().numpy(), want, rtol=0, atol=1e-9,
                                       err_msg=f"{tag}.{cls}")


# ---- 2-D gradients of the reference's own op sequence (dev/optimize_single_arc.py:31-47 pattern),
# including the NaN pattern of totally reflected rays (geometry.py:640-646)

GRAD2D_TAGS = ["arc", "seg", "grefr", "prism", "gtir"]
FLOAT2D = {"segments": GEO2, "arcs": ("x_center", "y_center", "angle_start", "angle_end", "radius")}


def _loss_2d(fin, act):
    """The scalar of make_reference_trace2d_golden.loss_of (rows of a 4 x n block or a dict)."""
    loss = 0.0
    if fin is not None:
        loss = loss + (fin[2] ** 2).sum() + 0.5 * (fin[3] * fin[0]).sum()
    if act is not None:
        loss = loss + 0.3 * act[3].sum()
    return loss


def _grad_scale(g, prefix):
    """Largest finite gradient magnitude over all fields of a scene (an entry the error does not
    depend on, e.g. a wall's end point moved along the wall, is rounding noise on both sides)."""
    vals = [np.abs(g[k][np.isfinite(g[k])]).max() for k in g.files
            if k.startswith(prefix) and np.isfinite(g[k]).any()]
    return max(vals) if vals else 1.0


def _check_grad(got, want, tag, key, rtol, scale=None):
    """Same non-finite entries; finite ones to rtol of `scale` (default: the field's largest
    finite magnitude)."""
    assert got.shape == want.shape, (tag, key)
    nan = ~np.isfinite(want)
    assert np.array_equal(~np.isfinite(got), nan), \
        f"{tag}.{key}: non-finite pattern {~np.isfinite(got)} vs reference {nan}"
    if (~nan).any():
        scale = scale or max(np.abs(want[~nan]).max(), 1e-300)
        err = np.abs(got[~nan] - want[~nan]).max() / scale
        assert err <= rtol, f"{tag}.{key}: rel err {err:.2e}"


@pytest.mark.parametrize("tag", GRAD2D_TAGS)
def test_oracle_2d_gradients_reproduce_the_reference_tape(tag):
    """torch.autograd through the oracle == torch.autograd through the reference's own source
    (the fixture), field by field, NaN for NaN: grefr / arc / seg have no total internal
    reflection (all finite); prism and gtir do (the entries the reflected rays touched are NaN,
    the others finite)."""
    from oracle import tracer
    g = np.load(TRACE2D)
    sets = _sets_2d(g, tag)
    leaves = []
    for name, fields in sets.items():
        for f in FLOAT2D[name.split("_")[1]]:
            fields[f] = fields[f].clone().requires_grad_(True)
            leaves.append((name, f, fields[f]))
    osys = tracer.System(2, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"],
                                       tracer.MATERIALS["reflective"]], **sets)
    rays, wl = g[tag + "_rays"], g[tag + "_wl"]
    src = {n: torch.tensor(rays[i]) for i, n in enumerate(GEO2)}
    src["wavelength"] = torch.tensor(wl)
    src["ray_id"] = torch.arange(rays.shape[1], dtype=torch.float64)
    ref = tracer.ray_trace(osys, src, max_iterations=4, inherit=("wavelength", "ray_id"),
                           flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    blk = lambda r: [r[n] for n in GEO2] if r else None
    loss = _loss_2d(blk(ref["finished"]), blk(ref["active"]))
    assert loss.item() == float(g[tag + "_loss"])
    got = torch.autograd.grad(loss, [t for _, _, t in leaves], allow_unused=True)
    n_nan = 0
    for (name, f, t), gr in zip(leaves, got):
        want = g[f"{tag}_grad__{name}__{f}"]
        have = np.zeros(t.shape) if gr is None else gr.numpy()
        _check_grad(have, want, tag, f"{name}.{f}", 1e-12, _grad_scale(g, f"{tag}_grad__"))
        n_nan += int(np.isnan(want).sum())
    assert (n_nan > 0) == (tag in ("prism", "gtir"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", GRAD2D_TAGS)
def test_hip_2d_gradients_reproduce_the_reference_tape(tag):
    """k_backward2d (float64 ray state) against the reference-source gradients: identical NaN
    pattern (default policy = the reference's: a totally reflected ray poisons what it touched),
    finite entries to 1e-8 of the field's largest gradient."""
    from tensorflowraytrace_amd import ops
    import test_gpu_trace2d as t2
    g = np.load(TRACE2D)
    sets = _sets_2d(g, tag)
    scene, seg, arc = t2._gpu_scene(sets, g[tag + "_wl"], requires_grad=True)
    src = torch.tensor(g[tag + "_rays"], dtype=torch.float64, device="cuda:0")
    out = ops.trace2d(src, scene, max_passes=4)
    fin = out["finished"].double() if out["finished"].shape[1] else None
    act = out["active"].double() if out["active"].shape[1] else None
    loss = _loss_2d(fin, act)
    np.testing.assert_allclose(loss.item(), float(g[tag + "_loss"]), rtol=1e-10)
    for kind, geo, cols in ((seg, GEO2, range(4)), (arc, FLOAT2D["arcs"], range(5))):
        if kind is None:
            continue
        (gr,) = torch.autograd.grad(loss, [kind["geo"]], retain_graph=True)
        gr = gr.cpu().numpy()
        suffix = "segments" if geo is GEO2 else "arcs"
        row = 0
        for cname in ("optical", "stop", "target"):        # merged order of _gpu_scene
            name = f"{cname}_{suffix}"
            if name not in sets:
                continue
            n = sets[name][geo[0]].shape[0]
            for c in cols:
                _check_grad(gr[row:row + n, c], g[f"{tag}_grad__{name}__{geo[c]}"], tag,
                            f"{name}.{geo[c]}", 1e-8, _grad_scale(g, f"{tag}_grad__"))
            row += n
        assert row == gr.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["prism", "gtir"])
def test_hip_2d_finite_tir_gradient_is_an_opt_in(tag):
    """tfrt_scene2d.finite_tir_gradient = 1: the reflect branch's own gradient, no NaN; equals
    the oracle asked for the same thing (oracle.geom.snells_law_2D(finite_tir_gradient=True))."""
    from oracle import tracer
    from tensorflowraytrace_amd import ops
    import test_gpu_trace2d as t2
    g = np.load(TRACE2D)
    sets = _sets_2d(g, tag)
    scene, seg, arc = t2._gpu_scene(sets, g[tag + "_wl"], requires_grad=True)
    scene.finite_tir_gradient = True
    src = torch.tensor(g[tag + "_rays"], dtype=torch.float64, device="cuda:0")
    out = ops.trace2d(src, scene, max_passes=4)
    loss = _loss_2d(out["finished"].double(), out["active"].double())
    kind, geo = (seg, GEO2) if seg is not None else (arc, FLOAT2D["arcs"])
    (gr,) = torch.autograd.grad(loss, [kind["geo"]])
    gr = gr.cpu().numpy()
    assert np.isfinite(gr).all()

    osets = _sets_2d(g, tag)
    name = "optical_segments" if seg is not None else "optical_arcs"
    leaves = []
    for f in geo:
        osets[name][f] = osets[name][f].clone().requires_grad_(True)
        leaves.append(osets[name][f])
    osys = tracer.System(2, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"],
                                       tracer.MATERIALS["reflective"]], **osets)
    rays = g[tag + "_rays"]
    osrc = {n: torch.tensor(rays[i]) for i, n in enumerate(GEO2)}
    osrc["wavelength"] = torch.tensor(g[tag + "_wl"])
    osrc["ray_id"] = torch.arange(rays.shape[1], dtype=torch.float64)
    ref = tracer.ray_trace(osys, osrc, max_iterations=4, inherit=("wavelength", "ray_id"),
                           finite_tir_gradient=True)
    rloss = _loss_2d([ref["finished"][n] for n in GEO2], [ref["active"][n] for n in GEO2])
    want = torch.autograd.grad(rloss, leaves, allow_unused=True)
    n = leaves[0].shape[0]
    for c, w in enumerate(want):
        w = np.zeros(n) if w is None else w.numpy()
        scale = max(np.abs(w).max(), 1e-300)
        assert np.abs(gr[:n, c] - w).max() / scale <= 1e-8, (tag, geo[c])


# ---------------------------------------------------------------------------------------------
# tests/golden/reference_soup3d.npz: 3-pass traces of adversarial triangle soups (coplanar ties,
# grazing rays, stops, targets, mirrors, 6 materials) by the reference's own engine, all five ray
# classes, with and without dead_ray_length (tests/golden/make_reference_soup_golden.py)

SOUP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_soup3d.npz")
CLASSES = ("finished", "active", "dead", "stopped", "unfinished")


def _soup_loss(fin, act, stp, L):
    """make_reference_soup_golden.soup_loss."""
    loss = 0.0
    if fin is not None and fin.shape[1]:
        loss = loss + ((fin[3] / L) ** 2).sum() + 0.5 * (fin[4] * fin[2]).sum() / L ** 2
    if act is not None and act.shape[1]:
        loss = loss + 0.3 * act[5].sum() / L
    if stp is not None and stp.shape[1]:
        loss = loss + 0.2 * stp[4].sum() / L
    return loss


def _soup_case(g, seed):
    n = torch.tensor(g["n_material"])
    mat_in, mat_out = torch.tensor(g[f"s{seed}__mat_in"]), torch.tensor(g[f"s{seed}__mat_out"])
    return dict(P=torch.tensor(g[f"s{seed}__P"]), cat=torch.tensor(g[f"s{seed}__cat"]),
                n_in=n[mat_in], n_out=n[mat_out], rays=torch.tensor(g[f"s{seed}__rays"]),
                L=float(g[f"s{seed}__L"]), dead=float(g[f"s{seed}__dead_ray_length"]))


@pytest.mark.parametrize("tag", ["plain", "deadlen"])
def test_oracle_reproduces_the_reference_engine_on_adversarial_soups(tag):
    from oracle import tracer
    g = np.load(SOUP)
    # (the dense oracle takes seconds per scene on a quiet host, a minute on a busy one: the
    # dead_ray_length variant skips the largest scene here; the GPU test below runs all of them)
    for seed in (g["seeds"] if tag == "plain" else g["seeds"][:3]):
        sc = _soup_case(g, seed)

        def sub(mask):
            verts = sc["P"][mask].reshape(-1, 3)
            d = tracer.faces_from_vertices(verts, torch.arange(verts.shape[0]).reshape(-1, 3))
            d["n_in"], d["n_out"] = sc["n_in"][mask], sc["n_out"][mask]
            return d
        cat = sc["cat"]
        if tag == "plain":
            sc["P"] = sc["P"].clone().requires_grad_(True)
        system = tracer.System(3, optical=sub(cat == 0), stop=sub(cat == 1), target=sub(cat == 2))
        src = {n: sc["rays"][i] for i, n in enumerate(NAMES)}
        src["ray_id"] = torch.arange(sc["rays"].shape[1], dtype=torch.float64)
        ref = tracer.ray_trace(system, src, max_iterations=int(g["passes"]), inherit=("ray_id",),
                               index_type="value", new_ray_length=sc["L"],
                               flags=dict(compile_dead_rays=True, compile_stopped_rays=True,
                                          dead_ray_length=sc["dead"] if tag == "deadlen" else None))
        for cls in CLASSES:
            want, want_id = g[f"s{seed}__{tag}__{cls}"], g[f"s{seed}__{tag}__{cls}_id"]
            rs = ref.get(cls) or {}
            n = rs["x_start"].shape[0] if "x_start" in rs else 0
            assert n == want.shape[1], (seed, cls)
            if n:
                assert np.array_equal(rs["ray_id"].numpy().astype(np.int64), want_id), (seed, cls)
                got = torch.stack([rs[k] for k in NAMES]).detach().numpy()
                assert np.array_equal(got, want), (seed, cls)                 # every bit
        if tag == "plain":
            # d loss / d face vertices: autograd over the oracle == autograd over the reference's
            # own op sequence (mirrors and total internal reflection included)
            blk = lambda rs: torch.stack([rs[k] for k in NAMES]) if rs else None
            loss = _soup_loss(blk(ref["finished"]), blk(ref["active"]), blk(ref["stopped"]), sc["L"])
            assert loss.item() == float(g[f"s{seed}__plain__loss"])
            (gP,) = torch.autograd.grad(loss, [sc["P"]])
            _check_grad(gP.numpy(), g[f"s{seed}__plain__grad_P"], f"soup {seed}", "P", 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "deadlen"])
def test_hip_reproduces_the_reference_engine_on_adversarial_soups(tag):
    """float64 ray state, hierarchy and all-pairs modes: every class holds the reference engine's
    rays, in its order, with every coordinate bit equal (coplanar ties, stops, dead rays cut to
    dead_ray_length, rays still travelling after the last pass)."""
    from tensorflowraytrace_amd import ops, _lib
    g = np.load(SOUP)
    dev = "cuda:0"
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    for seed in g["seeds"]:
        sc = _soup_case(g, seed)
        fv = sc["P"].to(dev).requires_grad_(tag == "plain")
        for clustered in (True, False):
            args = ops.Scene3DArgs(fv.detach(), sc["cat"].int().to(dev), n_in=sc["n_in"].to(dev),
                                   n_out=sc["n_out"].to(dev),
                                   cluster_order=ops.cluster_order(fv.detach()) if clustered else None)
            out = ops.trace3d(sc["rays"].to(dev), fv, args, max_passes=int(g["passes"]), flags=flags,
                              new_ray_length=sc["L"],
                              dead_ray_length=sc["dead"] if tag == "deadlen" else None)
            for cls in CLASSES:
                want, want_id = g[f"s{seed}__{tag}__{cls}"], g[f"s{seed}__{tag}__{cls}_id"]
                got = out[cls].detach().cpu().numpy()
                assert got.shape[1] == want.shape[1], (seed, cls, clustered)
                assert np.array_equal(out[cls + "_id"].cpu().numpy().astype(np.int64), want_id), \
                    (seed, cls, clustered)
                assert np.array_equal(got, want), (seed, cls, clustered)
            if tag == "plain":
                # the reverse sweep against the reference-source gradient of the same scalar
                loss = _soup_loss(out["finished"], out["active"], out["stopped"], sc["L"])
                (gP,) = torch.autograd.grad(loss, [fv])
                _check_grad(gP.cpu().numpy(), g[f"s{seed}__plain__grad_P"], f"soup {seed}", "P", 1e-8)
