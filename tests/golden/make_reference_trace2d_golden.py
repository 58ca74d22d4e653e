#!/usr/bin/env python3
"""
2-D counterpart of make_reference_trace_golden.py: the reference's own OpticalSystem2D /
OpticalEngine (tfrt/engine.py:254-866, 1544-1986: _segment_intersection, _arc_intersection,
_seg_or_arc, _get_arc_norm, process_projection_2D, single_pass, ray_trace) with
StandardReaction -> snells_law_2D, executed under tests/tf_shim on three scenes of
tests/test_gpu_trace2d.py::_scene: arcs only, segments only, and both.  Writes
tests/golden/reference_trace2d.npz (inputs and the finished / active / stopped / dead sets).

The mixed scene is the reference's mis-paired concat (engine.py:1958-1965): its output is kept as
evidence that the oracle's ``bug_compatible=True`` reproduces it; the product pairs rays and
boundary data correctly instead.

Gradients (the dev/optimize_single_arc.py:31-47 pattern: error of the finished rays, gradient
w.r.t. the boundary fields): torch.autograd through the reference's own OpticalSystem2D +
snells_law_2D op sequence, w.r.t. every float field of every boundary set, recorded WITH the
reference's non-finite entries -- ``tf.asin(theta2)`` in the unselected branch of
geometry.py:640-646 makes the gradient of every totally reflected ray NaN, which poisons each
boundary entry such a ray touched up to and including the reflecting one.  Scenes:
``grefr`` small convex arcs, refraction only (no NaN); ``arc`` and ``seg`` as above (no total internal reflection either: all radii of that seed are positive; a
mirror polyline); ``gtir`` arcs of alternating radius sign: total internal reflection near the
ends of the negative ones; ``prism`` acrylic bodies of segments: 45-degree total internal
reflection in a prism, plain refraction through a slab.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, os.path.join(TESTS, "tf_shim"))
if not os.path.exists("/root/reference/tfrt/engine.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import tfrt.engine as ref_engine        # noqa: E402  (the reference's modules)
import tfrt.materials as ref_materials  # noqa: E402
import tfrt.operation as ref_operation  # noqa: E402
from test_gpu_trace2d import _scene     # noqa: E402  (scene generator shared with the GPU tests)

GEO = ("x_start", "y_start", "x_end", "y_end")


class FieldSet(dict):
    dimension = 2

    def update(self):
        pass


FLOAT_FIELDS = {"segments": GEO, "arcs": ("x_center", "y_center", "angle_start", "angle_end", "radius")}


def loss_of(eng):
    """A generic scalar of the finished and the active-history rays (both are concatenations over
    all passes, so every pass but the last has its Snell step on the gradient path)."""
    fin, act = eng.finished_rays, eng.active_rays
    loss = torch.zeros((), dtype=torch.float64)
    if bool(fin):
        loss = loss + (fin["x_end"] ** 2).sum() + 0.5 * (fin["y_end"] * fin["x_start"]).sum()
    if bool(act):
        loss = loss + 0.3 * act["y_end"].sum()
    return loss


def trace(sets, rays, wl, passes, grads=False):
    system = ref_engine.OpticalSystem2D()
    leaves = []
    if grads:
        sets = {name: dict(fields) for name, fields in sets.items()}
        for name, fields in sets.items():
            for f in FLOAT_FIELDS[name.split("_")[1]]:
                fields[f] = fields[f].clone().requires_grad_(True)
                leaves.append((name, f, fields[f]))
    for name, fields in sets.items():
        setattr(system, name, [FieldSet(fields)])
    src = FieldSet({k: torch.tensor(rays[i]) for i, k in enumerate(GEO)})
    src["wavelength"] = torch.tensor(wl)
    src["ray_id"] = torch.arange(rays.shape[1], dtype=torch.float64)
    system.sources = [src]
    system.materials = [{"n": ref_materials.vacuum}, {"n": ref_materials.acrylic},
                        {"n": ref_materials.reflective}]
    system.update()
    eng = ref_engine.OpticalEngine(
        2, [ref_operation.StandardReaction()], compile_dead_rays=True, compile_stopped_rays=True,
        simple_ray_inheritance={"wavelength", "ray_id"})
    eng.optical_system = system
    eng.validate_system()
    eng.ray_trace(passes)
    out = {}
    if grads:
        loss = loss_of(eng)
        got = torch.autograd.grad(loss, [t for _, _, t in leaves], allow_unused=True)
        out["loss"] = np.float64(loss.item())
        for (name, f, t), g in zip(leaves, got):
            # (a field the error does not depend on -- the arcs' angles only feed comparisons --
            # has no gradient in the tape: recorded as zeros)
            out[f"grad__{name}__{f}"] = np.zeros(t.shape) if g is None else g.numpy()
    for cls, rs in (("finished", eng.finished_rays), ("active", eng.active_rays),
                    ("stopped", eng.stopped_rays), ("dead", eng.dead_rays)):
        if bool(rs):
            out[cls] = torch.stack([rs[g] for g in GEO]).detach().numpy()
            out[cls + "_id"] = rs["ray_id"].detach().numpy().astype(np.int64)
        else:
            out[cls] = np.zeros((4, 0))
            out[cls + "_id"] = np.zeros(0, dtype=np.int64)
    return out


PI = np.pi


def refract_scene(rng, n_rays):
    """Small convex acrylic arcs that do not overlap, hit from outside only: every ray refracts
    once (or misses) and runs on to the target arc -- no total internal reflection anywhere."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    k = 6
    sets = {"optical_arcs": dict(
        x_center=t(np.linspace(-2.5, 2.5, k) + rng.normal(size=k) * 0.05),
        y_center=t(rng.normal(size=k) * 0.1 + 3.0), angle_start=t(np.full(k, -PI + 0.3)),
        angle_end=t(np.full(k, -0.3)), radius=t(rng.uniform(0.3, 0.45, k)),
        mat_in=torch.ones(k, dtype=torch.int64), mat_out=torch.zeros(k, dtype=torch.int64)),
        "target_arcs": dict(x_center=t([0.0]), y_center=t([0.0]), angle_start=t([0.2]),
                            angle_end=t([PI - 0.2]), radius=t([9.0]))}
    ang = rng.uniform(0.4 * PI, 0.6 * PI, n_rays)
    x0 = rng.uniform(-3, 3, n_rays)
    rays = np.stack([x0, np.zeros(n_rays), x0 + np.cos(ang), np.sin(ang)])
    return sets, rays, rng.uniform(450, 650, n_rays)


def arc_tir_scene(rng, n_rays):
    """The arcs of test_gpu_trace2d._scene with alternating radius signs: a negative radius turns
    the norm towards the centre, rays from below then meet the arc from the acrylic side
    (geometry.py:596-600 "internal") and are totally reflected near its ends."""
    sets, rays, wl = _scene(rng, n_rays, with_seg=False, with_arc=True)
    # (0.35 x the radii: the circles no longer overlap, so some arcs never see a reflected ray)
    r = sets["optical_arcs"]["radius"].abs() * 0.35
    sets["optical_arcs"]["radius"] = r * torch.tensor([1.0, -1.0, 1.0, 1.0, -1.0, 1.0], dtype=torch.float64)
    return sets, rays, wl


def prism_scene(rng, n_rays):
    """Two acrylic bodies made of segments (norms point out of the glass: mat_in = acrylic).
    A: right-angle prism (0,0)-(2,0)-(0,2); rays enter through the bottom leg heading up and meet
    the hypotenuse at 45 +- 8 degrees (critical angle 42.2): most are totally reflected and leave
    through the left leg towards the left target wall, the rest refract out towards the top wall.
    B: a slab at x in [6, 8]: plain refraction in and out."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    jit = lambda: rng.normal(size=2) * 0.01
    a0, a1, a2 = np.array([0.0, 0.0]) + jit(), np.array([2.0, 0.0]) + jit(), np.array([0.0, 2.0]) + jit()
    b = [np.array(p) + jit() for p in ((6.0, 1.0), (8.0, 1.1), (8.0, 1.9), (6.0, 2.0))]
    segs = [(a1, a0), (a0, a2), (a2, a1),                  # bottom leg, left leg, hypotenuse
            (b[1], b[0]), (b[0], b[3]), (b[3], b[2]), (b[2], b[1])]
    opt = np.array([[p[0], p[1], q[0], q[1]] for p, q in segs])
    sets = {"optical_segments": dict(
        x_start=t(opt[:, 0]), y_start=t(opt[:, 1]), x_end=t(opt[:, 2]), y_end=t(opt[:, 3]),
        mat_in=torch.ones(len(segs), dtype=torch.int64),
        mat_out=torch.zeros(len(segs), dtype=torch.int64)),
        "target_segments": dict(x_start=t([-5.0, -6.0]), y_start=t([-3.0, 9.0]),
                                x_end=t([-5.0, 14.0]), y_end=t([9.0, 9.2]))}
    half = n_rays // 2
    x0 = np.concatenate([rng.uniform(0.15, 1.7, half), rng.uniform(6.6, 7.4, n_rays - half)])
    ang = 0.5 * PI + rng.uniform(-0.21, 0.21, n_rays)
    rays = np.stack([x0, np.full(n_rays, -1.0), x0 + np.cos(ang), -1.0 + np.sin(ang)])
    return sets, rays, rng.uniform(450, 650, n_rays)


def main():
    doc = {}
    scenes = []
    for tag, seed, with_seg, with_arc in (("arc", 11, False, True), ("seg", 12, True, False),
                                          ("both", 13, True, True)):
        rng = np.random.default_rng(seed)
        scenes.append((tag, *_scene(rng, 1500, with_seg=with_seg, with_arc=with_arc)))
    scenes.append(("grefr", *refract_scene(np.random.default_rng(21), 600)))
    scenes.append(("prism", *prism_scene(np.random.default_rng(22), 600)))
    scenes.append(("gtir", *arc_tir_scene(np.random.default_rng(23), 600)))
    for tag, sets, rays, wl in scenes:
        # (the mixed scene's output is the reference's mis-paired concat: no gradients of that)
        out = trace(sets, rays, wl, 4, grads=tag != "both")
        doc[f"{tag}_rays"], doc[f"{tag}_wl"] = rays, wl
        for name, fields in sets.items():
            for f, v in fields.items():
                doc[f"{tag}__{name}__{f}"] = v.numpy()
        for k, v in out.items():
            doc[f"{tag}_{k}"] = v
        print(tag, {k: v.shape for k, v in out.items() if not k.endswith("_id") and "grad" not in k},
              {k.split("__", 1)[1]: f"{int(np.isnan(v).sum())} NaN of {v.size}"
               for k, v in out.items() if k.startswith("grad__")})
    np.savez_compressed(os.path.join(HERE, "reference_trace2d.npz"), **doc)


if __name__ == "__main__":
    main()
