#!/usr/bin/env python3
"""
Golden vectors of the 3-D trace produced by EXECUTING the reference's own engine
(tfrt/engine.py: OpticalSystem3D._merge_boundaries / intersect / _intersection,
OpticalEngine.process_projection_3D / single_pass / ray_trace; tfrt/operation.py
StandardReaction.main; tfrt/materials.py; tfrt/geometry.py) in the build container under the
TensorFlow stand-in of tests/tf_shim.  Writes tests/golden/reference_trace3d.npz.

The scene is tests/scene_util.lens_scene (two hex-mesh acrylic surfaces + target).  Boundaries and
the source enter the reference engine as plain field sets (tfrt/boundaries.py itself needs pyvista,
which is not installed): the face fields xp..z2 and norm are formed here with the formula of
boundaries.py:890-923, from vertices = zero + p * (1,0,0).  Because the stand-in's ops are torch
ops, torch.autograd through the reference's op sequence gives the gradients tf.GradientTape would
(tf.where / gather semantics are the same): d loss / d p for both surfaces is stored as well.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.join(TESTS, "tf_shim"))
if not os.path.exists("/root/reference/tfrt/engine.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import scene_util                      # noqa: E402
import tfrt.engine as ref_engine       # noqa: E402  (the reference's modules)
import tfrt.materials as ref_materials  # noqa: E402
import tfrt.operation as ref_operation  # noqa: E402

F64 = torch.float64


class FieldSet(dict):
    """What the reference's systems need of a boundary / source: a dict of fields, a dimension
    and an update() to register."""
    dimension = 3

    def update(self):
        pass


def faces_from_vertices(verts, faces):
    """boundaries.py:890-923 with the stand-in's arithmetic (products and differences rounded one
    by one, correctly rounded sqrt)."""
    import tensorflow as tf
    faces = torch.as_tensor(faces, dtype=torch.int64)
    first, second, third = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    a, b = second - first, third - second
    cross = torch.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                         a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], dim=1)
    from oracle import geom            # its correctly rounded, differentiable sqrt
    norm = cross / geom.sqrt(torch.sum(cross * cross, dim=1, keepdim=True))
    out = FieldSet(norm=norm)
    for name, pts in zip(("p", "1", "2"), (first, second, third)):
        out["x" + name], out["y" + name], out["z" + name] = pts[:, 0], pts[:, 1], pts[:, 2]
    return out


def main():
    sys.path.insert(0, os.path.dirname(TESTS))
    n_rays, passes = 1200, 4
    scene = scene_util.lens_scene(n_rays, k_front=5, k_back=4)
    tt = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    p_f = tt(scene["p_f"]).requires_grad_(True)
    p_b = tt(scene["p_b"]).requires_grad_(True)
    vec = tt(scene["vector"]).reshape(1, 3)
    front = faces_from_vertices(tt(scene["zero_f"]) + p_f.reshape(-1, 1) * vec, scene["faces_f"])
    back = faces_from_vertices(tt(scene["zero_b"]) + p_b.reshape(-1, 1) * vec, scene["faces_b"])
    for s in (front, back):
        n = s["xp"].shape[0]
        s["mat_in"] = torch.ones(n, dtype=torch.int64)
        s["mat_out"] = torch.zeros(n, dtype=torch.int64)
    target = faces_from_vertices(tt(scene["target_verts"]), scene["target_faces"])
    rays = scene["rays"].astype(np.float32).astype(np.float64)      # what a float32-state trace sees
    source = FieldSet({k: tt(rays[i]) for i, k in enumerate(
        ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"))})
    source["wavelength"] = tt(scene["wavelength"])
    source["ray_id"] = torch.arange(n_rays, dtype=F64)

    system = ref_engine.OpticalSystem3D()
    system.optical = [front, back]
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": ref_materials.vacuum}, {"n": ref_materials.acrylic}]
    system.update()
    eng = ref_engine.OpticalEngine(
        3, [ref_operation.StandardReaction()], compile_dead_rays=True, compile_stopped_rays=True,
        simple_ray_inheritance={"wavelength", "ray_id"})
    eng.optical_system = system
    eng.validate_system()
    eng.ray_trace(passes)

    geo = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
    out = {k: scene[k] for k in ("zero_f", "faces_f", "p_f", "zero_b", "faces_b", "p_b", "vector",
                                 "target_verts", "target_faces", "wavelength", "goal")}
    out["rays"] = rays
    out["passes"] = np.int64(passes)
    for cls, rs in (("finished", eng.finished_rays), ("active", eng.active_rays), ("dead", eng.dead_rays)):
        out[cls] = torch.stack([rs[g] for g in geo]).detach().numpy()
        out[cls + "_id"] = rs["ray_id"].detach().numpy().astype(np.int64)
    fin = eng.finished_rays
    goal = tt(scene["goal"])[fin["ray_id"].long()]
    loss = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
    g_f, g_b = torch.autograd.grad(loss, [p_f, p_b])
    out["loss"], out["grad_front"], out["grad_back"] = loss.item(), g_f.numpy(), g_b.numpy()
    np.savez_compressed(os.path.join(HERE, "reference_trace3d.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()
           if k in ("finished", "active", "dead", "loss")})


if __name__ == "__main__":
    main()
