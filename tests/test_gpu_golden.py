"""HIP path against the committed golden vectors (no oracle at run time)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-9), ("f32", torch.float32, 1e-5)])
def test_lens3d_golden(tag, dtype, tol):
    from tensorflowraytrace_amd import ops, _lib
    from test_gpu_trace3d import _gpu_scene
    g = np.load(os.path.join(GOLD, "lens3d.npz"))
    scene = {k: g[k] for k in ("zero_f", "faces_f", "p_f", "zero_b", "faces_b", "p_b", "vector",
                               "target_verts", "target_faces", "rays", "wavelength", "goal")}
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    out = ops.trace3d(src, fv, sc, max_passes=4, flags=flags)
    for cls in ("finished", "active", "dead"):
        want = g[f"{tag}_{cls}"]
        assert np.array_equal(out[cls + "_id"].cpu().numpy(), g[f"{tag}_{cls}_id"])
        if want.size:
            err = np.abs(out[cls].detach().cpu().double().numpy() - want).max() / max(1.0, np.abs(want).max())
            assert err <= tol, (cls, err)
    fin = out["finished"]
    goal = torch.tensor(scene["goal"], device=fin.device)[out["finished_id"].long()]
    err = ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
    g_f, g_b = torch.autograd.grad(err, [p_f, p_b])
    assert abs(err.item() - float(g[f"{tag}_error"])) <= 10 * tol * float(g[f"{tag}_error"])
    for got, want in ((g_f, g[f"{tag}_grad_front"]), (g_b, g[f"{tag}_grad_back"])):
        rel = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        assert rel <= max(tol, 1e-8), rel


def test_geometry_golden():
    from tensorflowraytrace_amd import ops
    g = np.load(os.path.join(GOLD, "geometry.npz"))
    dev = "cuda:0"
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    s, e = g["tri_s"], g["tri_e"]
    out = ops.snell3d(*[t(s[:, i]) for i in range(3)], *[t(e[:, i]) for i in range(3)],
                      t(g["sn_norm"]), t(g["sn_n_in"]), t(g["sn_n_out"]), 1.25)
    np.testing.assert_allclose(out.cpu().numpy(), g["sn3"], rtol=0, atol=1e-13)
    out = ops.snell2d(t(s[:, 0]), t(s[:, 1]), t(e[:, 0]), t(e[:, 1]), t(g["sn2_norm"]),
                      t(g["sn_n_in"]), t(g["sn_n_out"]), 0.75)
    np.testing.assert_allclose(out.cpu().numpy(), g["sn2"], rtol=0, atol=1e-12)
    # one ray against one triangle at a time == the 1:1 golden
    P9 = g["tri_P"]
    hits = 0
    for i in range(0, s.shape[0], 9):
        rays = t(np.concatenate([s[i], e[i]]).reshape(6, 1))
        x, y, z, valid, ray_u, tu, tv, gi = ops.intersect3d(rays, t(P9[i:i + 1]), 1e-10, 1e300, -1e300)
        if g["tri_valid"][i]:
            hits += 1
            assert bool(valid[0])
            np.testing.assert_allclose([float(ray_u[0]), float(tu[0]), float(tv[0])],
                                       [g["tri_ray_u"][i], g["tri_u"][i], g["tri_v"][i]], rtol=1e-13)
    assert hits > 5


def test_scene2d_golden():
    from tensorflowraytrace_amd import ops, _lib
    from oracle import tracer  # material formulas only (n(lambda) table), not the checker
    g = np.load(os.path.join(GOLD, "scene2d.npz"))
    dev = "cuda:0"
    t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    seg_geo = torch.cat([
        torch.stack([t(g["seg_" + k]) for k in ("x_start", "y_start", "x_end", "y_end")], 1),
        torch.stack([t(g["wall_" + k]) for k in ("x_start", "y_start", "x_end", "y_end")], 1)]).requires_grad_(True)
    arc_geo = torch.stack([t(g["arc_" + k]) for k in ("x_center", "y_center", "angle_start", "angle_end", "radius")], 1).requires_grad_(True)
    i32 = torch.int32
    seg = dict(geo=seg_geo, cat=t([0] * 6 + [2] * 2, i32), mat_in=t([2] * 6 + [0] * 2, i32),
               mat_out=t([0] * 8, i32), n_in=None, n_out=None)
    arc = dict(geo=arc_geo, cat=t([0] * 3, i32), mat_in=t([1] * 3, i32), mat_out=t([0] * 3, i32),
               n_in=None, n_out=None)
    wl = torch.tensor(g["wavelength"])
    n_table = torch.stack([tracer.MATERIALS[m](wl) for m in ("vacuum", "acrylic", "reflective")]).to(dev)
    # (scene2d.npz holds the gradients of the oracle's finite form of total internal reflection --
    # every arc of this scene reflects some ray, so the reference's own gradient is NaN throughout;
    # that policy is pinned by tests/golden/reference_trace2d.npz in test_reference_golden.py)
    scene = ops.Scene2DArgs(seg, arc, n_table, True, False, finite_tir_gradient=True)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    out = ops.trace2d(t(g["rays"]), scene, 5, flags=flags)
    for cls in ("finished", "active", "dead"):
        assert np.array_equal(out[cls + "_id"].cpu().numpy(), g[cls + "_id"])
        if g[cls].size:
            np.testing.assert_allclose(out[cls].detach().cpu().numpy(), g[cls], rtol=0, atol=1e-9)
    loss = (out["finished"][3] ** 2).sum() + 0.3 * out["active"][3].sum()
    gs, ga = torch.autograd.grad(loss, [seg_geo, arc_geo])
    assert abs(loss.item() - float(g["loss"])) < 1e-9 * float(g["loss"])
    np.testing.assert_allclose(gs.cpu().numpy()[:6], g["grad_seg"], rtol=0,
                               atol=1e-8 * np.abs(g["grad_seg"]).max())
    np.testing.assert_allclose(ga.cpu().numpy()[:, [0, 1, 4]], g["grad_arc"], rtol=0,
                               atol=1e-8 * np.abs(g["grad_arc"]).max())
