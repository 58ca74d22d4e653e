"""
Generates the committed golden vectors under tests/golden/ from the float64 oracle.

    python tests/golden/make_golden.py

TensorFlow is not installed in the build container, so the vectors cannot come from the
reference itself; they are produced by the oracle, which is pinned by
the reference's own test properties (tests/test_oracle_reference_properties.py) and by the
analytic cases (tests/test_oracle_analytic.py).  A fixture is data only: inputs and expected
outputs.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import scene_util  # noqa: E402
import oracle_util  # noqa: E402
from oracle import geom, tracer  # noqa: E402

PI = math.pi


def lens3d():
    scene = scene_util.lens_scene(400, k_front=3, k_back=2, seed=11)
    out = {k: v for k, v in scene.items()}
    for tag, dt in (("f64", None), ("f32", np.float32)):
        system, (p_f, p_b), _ = oracle_util.lens_oracle(scene)
        ref = tracer.ray_trace(system, oracle_util.source_dict(scene["rays"], scene["wavelength"], dt),
                               max_iterations=4, inherit=("wavelength", "ray_id"),
                               flags=dict(compile_dead_rays=True))
        fin = ref["finished"]
        goal = torch.tensor(scene["goal"])[fin["ray_id"].long()]
        err = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
        g_f, g_b = torch.autograd.grad(err, [p_f, p_b])
        for cls in ("finished", "active", "dead"):
            out[f"{tag}_{cls}"] = oracle_util.block(ref[cls])
            out[f"{tag}_{cls}_id"] = (ref[cls]["ray_id"].numpy().astype(np.int32)
                                      if ref[cls] else np.zeros(0, np.int32))
        out[f"{tag}_error"] = float(err)
        out[f"{tag}_grad_front"] = g_f.numpy()
        out[f"{tag}_grad_back"] = g_b.numpy()
    np.savez_compressed(os.path.join(HERE, "lens3d.npz"), **out)


def scene2d():
    rng = np.random.default_rng(21)
    n = 300
    arcs = dict(x_center=np.array([-1.5, 0.0, 1.5]), y_center=np.array([3.0, 3.1, 2.9]),
                angle_start=np.full(3, -PI + 0.3), angle_end=np.full(3, -0.3),
                radius=np.array([1.0, -1.2, 0.9]))
    xs = np.linspace(-4, 4, 7)
    ys = 5.0 + 0.3 * np.sin(xs)
    segs = dict(x_start=xs[:-1], y_start=ys[:-1], x_end=xs[1:], y_end=ys[1:])
    walls = dict(x_start=np.array([6.0, -6.0]), y_start=np.array([-1.0, 8.0]),
                 x_end=np.array([6.0, -6.0]), y_end=np.array([8.0, -1.0]))
    ang = rng.uniform(0.25 * PI, 0.75 * PI, n)
    x0 = rng.uniform(-3, 3, n)
    rays = np.stack([x0, np.zeros(n), x0 + np.cos(ang), np.sin(ang)])
    wl = rng.uniform(450, 650, n)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    oa = {k: t(v).requires_grad_(k in ("x_center", "y_center", "radius")) for k, v in arcs.items()}
    oa["mat_in"] = torch.ones(3, dtype=torch.int64)
    oa["mat_out"] = torch.zeros(3, dtype=torch.int64)
    os_ = {k: t(v).requires_grad_(True) for k, v in segs.items()}
    os_["mat_in"] = torch.full((6,), 2, dtype=torch.int64)
    os_["mat_out"] = torch.zeros(6, dtype=torch.int64)
    system = tracer.System(
        2, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"], tracer.MATERIALS["reflective"]],
        optical_arcs=oa, optical_segments=os_, target_segments={k: t(v) for k, v in walls.items()})
    src = {nme: t(rays[i]) for i, nme in enumerate(("x_start", "y_start", "x_end", "y_end"))}
    src["wavelength"] = t(wl)
    src["ray_id"] = torch.arange(n, dtype=torch.float64)
    # (the finite form of the total-internal-reflection gradient: the reference's own is NaN for
    # every arc of this scene, see oracle.geom.snells_law_2D and tests/golden/reference_trace2d.npz)
    ref = tracer.ray_trace(system, src, max_iterations=5, inherit=("wavelength", "ray_id"),
                           flags=dict(compile_dead_rays=True), finite_tir_gradient=True)
    loss = (ref["finished"]["y_end"] ** 2).sum() + 0.3 * ref["active"]["y_end"].sum()
    leaves = [oa["x_center"], oa["y_center"], oa["radius"]] + [os_[k] for k in ("x_start", "y_start", "x_end", "y_end")]
    grads = torch.autograd.grad(loss, leaves)
    out = dict(rays=rays, wavelength=wl, loss=float(loss))
    out.update({"arc_" + k: v for k, v in arcs.items()})
    out.update({"seg_" + k: v for k, v in segs.items()})
    out.update({"wall_" + k: v for k, v in walls.items()})
    for cls in ("finished", "active", "dead"):
        out[cls] = oracle_util.block(ref[cls], dim=2)
        out[cls + "_id"] = ref[cls]["ray_id"].numpy().astype(np.int32) if ref[cls] else np.zeros(0, np.int32)
    out["grad_arc"] = np.stack([g.numpy() for g in grads[:3]], 1)
    out["grad_seg"] = np.stack([g.numpy() for g in grads[3:]], 1)
    np.savez_compressed(os.path.join(HERE, "scene2d.npz"), **out)


def geometry_vectors():
    rng = np.random.default_rng(31)
    n = 200
    out = {}
    P9 = rng.normal(size=(n, 9))
    s = rng.normal(size=(n, 3)) * 2
    e = rng.normal(size=(n, 3)) * 2
    e[:5] = s[:5]                       # degenerate rays
    P9[5:10, 6:9] = P9[5:10, 3:6]       # degenerate triangles
    r = geom.raw_line_triangle_intersect(*s.T, *e.T, *P9.T, 1e-10)
    out.update(tri_s=s, tri_e=e, tri_P=P9, tri_x=r[0].numpy(), tri_y=r[1].numpy(), tri_z=r[2].numpy(),
               tri_valid=r[3].numpy(), tri_ray_u=r[4].numpy(), tri_u=r[5].numpy(), tri_v=r[6].numpy())
    norm = rng.normal(size=(n, 3))
    n_in = rng.choice([0.0, 1.0, 1.33, 1.5], n)
    n_out = rng.choice([1.0, 1.2, 1.5], n)
    o = geom.snells_law_3D(*s.T, *e.T, norm, n_in, n_out, 1.25)
    out.update(sn_norm=norm, sn_n_in=n_in, sn_n_out=n_out, sn3=np.stack([v.numpy() for v in o]))
    na = rng.uniform(-2 * PI, 2 * PI, n)
    o = geom.snells_law_2D(s[:, 0], s[:, 1], e[:, 0], e[:, 1], na, n_in, n_out, 0.75)
    out.update(sn2_norm=na, sn2=np.stack([v.numpy() for v in o]))
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)


if __name__ == "__main__":
    lens3d()
    scene2d()
    geometry_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
