#!/usr/bin/env python3
"""
Golden vectors produced by EXECUTING the reference's own tfrt/geometry.py (build container only:
/root/reference does not exist on the GPU box) under the minimal TensorFlow stand-in of
tests/tf_shim (torch-CPU float64, one correctly rounded op per tf op).  Writes
tests/golden/reference_geometry.npz: seeded random inputs and the outputs of

    raw_line_intersect, raw_line_triangle_intersect, raw_line_circle_intersect,
    angle_in_interval, snells_law_2D, snells_law_3D        (tfrt/geometry.py:96-802)

The fixture holds data only.  tests/test_reference_golden.py then checks the oracle (CPU) and the
HIP entry points (GPU) against it.  What this pins and what not: the reference's operation ORDER,
masking and formulas as written in its source; not TensorFlow's own kernels (the stand-in supplies
the arithmetic), so the evidence is "restatement transcribed correctly", not "TensorFlow ran".
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tfrt/geometry.py"
PI = np.pi


def load_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tf_shim"))
    spec = importlib.util.spec_from_file_location("reference_geometry", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def inputs(seed=20261004):
    rng = np.random.default_rng(seed)
    n = 1500
    d = {}
    # line x line (element-wise), incl. exactly parallel pairs
    a = rng.normal(size=(8, n)) * 3
    a[6:8, :100] = a[4:6, :100] + (a[2:4, :100] - a[0:2, :100]) * rng.uniform(0.5, 2, 100)   # parallel
    d["li"] = a
    # line x triangle (element-wise): rays aimed at a random point near the triangle
    P = rng.normal(size=(9, n)) * 10 ** rng.uniform(-1, 1, n)
    w = rng.dirichlet((1, 1, 1), n).T * rng.uniform(0.5, 1.6, n)
    hit = P[0:3] * w[0] + P[3:6] * w[1] + P[6:9] * w[2]
    s = hit + rng.normal(size=(3, n)) * 3
    e = s + (hit - s) * rng.uniform(0.3, 2.5, n)
    P[:, :100] = P[:, 100:200]
    s[:, :50] = 0.0
    e[:, :50] = P[3:6, :50] - P[0:3, :50]          # rays parallel to an edge through the origin
    d["tri_rays"], d["tri"] = np.concatenate([s, e]), P
    # line x circle: through, tangent-ish, missing
    c = rng.normal(size=(2, n)) * 2
    r = rng.uniform(0.2, 3.0, n) * rng.choice([-1.0, 1.0], n)
    off = rng.uniform(-1.3, 1.3, n) * np.abs(r)
    off[:150] = np.abs(r[:150]) * (1 + rng.uniform(-1e-12, 1e-12, 150))                  # grazing
    th = rng.uniform(0, 2 * PI, n)
    foot = c + off * np.stack([np.cos(th), np.sin(th)])
    dirn = np.stack([-np.sin(th), np.cos(th)])
    ls = foot - dirn * rng.uniform(0.5, 4, n)
    le = ls + dirn * rng.uniform(0.2, 3, n)
    d["circ_lines"], d["circ"] = np.concatenate([ls, le]), np.concatenate([c, r[None]])
    # angle_in_interval
    d["ang"] = np.stack([rng.uniform(-2 * PI, 2 * PI, n), rng.uniform(-PI, PI, n), rng.uniform(-PI, PI, n)])
    # Snell 2-D / 3-D incl. TIR, mirrors (n_in = 0), n_out = 0
    n_in = rng.uniform(1.0, 1.8, n)
    n_out = rng.uniform(1.0, 1.8, n)
    n_in[:300] = 0.0
    n_out[300:500] = 0.0
    d["sn_n"] = np.stack([n_in, n_out])
    s2 = rng.normal(size=(2, n)) * 2
    d["sn2_rays"] = np.concatenate([s2, s2 + rng.normal(size=(2, n))])
    d["sn2_norm"] = rng.uniform(-2 * PI, 2 * PI, n)
    s3 = rng.normal(size=(3, n)) * 2
    d["sn3_rays"] = np.concatenate([s3, s3 + rng.normal(size=(3, n))])
    d["sn3_norm"] = rng.normal(size=(n, 3)) * 10 ** rng.uniform(-2, 2, (n, 1))
    return d


def run(mod, d):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
    out = {}
    eps = t(1e-10)
    x, y, valid, u, v = mod.raw_line_intersect(*[t(r) for r in d["li"]], eps)
    out["li_out"] = torch.stack([x, y, u, v]).numpy()
    out["li_valid"] = valid.numpy()
    res = mod.raw_line_triangle_intersect(*[t(r) for r in d["tri_rays"]], *[t(r) for r in d["tri"]], eps)
    out["tri_out"] = torch.stack([res[0], res[1], res[2], res[4], res[5], res[6]]).numpy()
    out["tri_valid"] = res[3].numpy()
    plus, minus = mod.raw_line_circle_intersect(*[t(r) for r in d["circ_lines"]], *[t(r) for r in d["circ"]], eps)
    for name, root in (("plus", plus), ("minus", minus)):
        out[f"circ_{name}"] = torch.stack([root["x"], root["y"], root["u"], root["v"]]).numpy()
        out[f"circ_{name}_valid"] = root["valid"].numpy()
    out["ang_out"] = mod.angle_in_interval(*[t(r) for r in d["ang"]]).numpy()
    L = 0.37
    o2 = mod.snells_law_2D(*[t(r) for r in d["sn2_rays"]], t(d["sn2_norm"]), t(d["sn_n"][0]), t(d["sn_n"][1]), L)
    out["sn2_out"] = torch.stack(list(o2)).numpy()
    o3 = mod.snells_law_3D(*[t(r) for r in d["sn3_rays"]], t(d["sn3_norm"]), t(d["sn_n"][0]), t(d["sn_n"][1]), L)
    out["sn3_out"] = torch.stack(list(o3)).numpy()
    out["new_ray_length"] = np.float64(L)
    return out


if __name__ == "__main__":
    if not os.path.exists(REF):
        raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
    mod = load_reference()
    d = inputs()
    out = run(mod, d)
    np.savez_compressed(os.path.join(HERE, "reference_geometry.npz"), **d, **out)
    print({k: v.shape for k, v in out.items() if hasattr(v, "shape")})
