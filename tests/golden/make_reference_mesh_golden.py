#!/usr/bin/env python3
"""
Golden vectors of the mesh / parametrisation tooling and of the
cylindrical light guide, from the reference's OWN tfrt/mesh_tools.py and tfrt/boundaries.py
executed under tests/tf_shim (its pyvista placeholder carries points / faces only, which is all
these functions touch).  Writes tests/golden/reference_mesh.npz:

* hexagonal_mesh, circular_mesh, cylindrical_mesh (mesh_tools.py:576-952): points and faces
* mesh_parametrization_tools (vertex_update_map + ancestor accumulator, :221-331),
  mesh_smoothing_tool (:345-421), find_generations,
  get_closest_point, get_flat_initial, gaussian_weights
* ParametricCylindricalGuide without caps (boundaries.py:1416-1617), rotationally symmetric and
  per-vertex: parameters after the p -= min(p) constraint, face fields, and the gradient of a fixed
  linear functional of the fields w.r.t. the parameters (repeat + vertex_update_map reverse).
  (With caps the reference drops the cap vertices from its vertex array but keeps face indices
  that count them, boundaries.py:1607-1611 -- a one-vertex shift this package does not reproduce.)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.join(TESTS, "tf_shim"))
if not os.path.exists("/root/reference/tfrt/mesh_tools.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import tfrt.boundaries as B      # noqa: E402  (the reference's modules)
import tfrt.mesh_tools as MT     # noqa: E402

TRI = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2")


class Point(np.ndarray):
    """A numpy vector that also takes part in arithmetic with the stand-in's tensors the way it
    would with tf tensors (``ndarray - tf.Tensor`` is a tf.Tensor): the guide hands its end points
    both to numpy code (cylindrical_mesh) and to FromAxisVG (``point - axis_point``)."""

    def __new__(cls, values):
        return np.asarray(values, dtype=np.float64).view(cls)

    def __sub__(self, other):
        if isinstance(other, torch.Tensor):
            return torch.as_tensor(np.asarray(self)) - other
        return np.asarray(self) - other


def main():
    rng = np.random.default_rng(3)
    doc = {}
    h = MT.hexagonal_mesh(radius=1.0, step_count=4)
    doc["hex_points"], doc["hex_faces"] = h.points, h.faces
    c = MT.circular_mesh(1.0, 0.3)
    doc["circ_points"], doc["circ_faces"] = c.points, c.faces
    cyl = MT.cylindrical_mesh((0.0, 0.0, 0.0), (0.0, 0.0, 3.0), radius=0.5, theta_res=8, z_res=5,
                              start_cap=True, end_cap=True)
    doc["cyl_points"], doc["cyl_faces"] = cyl.points, cyl.faces
    top = MT.get_closest_point(h, (0.0, 0.0, 0.0))
    doc["hex_top"] = np.int64(top)
    vmap, acc = MT.mesh_parametrization_tools(h, top)
    doc["hex_vmap"], doc["hex_acc"] = np.asarray(vmap), np.asarray(acc, dtype=np.float64)
    doc["hex_smoother"] = np.asarray(MT.mesh_smoothing_tool(h, [4, 2, 1]), dtype=np.float64)
    # (gradient_accumulator_1p raises NameError at the reference's HEAD: mesh_tools.py:64 calls an
    # undefined get_unique_edges)
    gens = MT.find_generations(top, h)
    doc["hex_generation_sizes"] = np.array([len(x) for x in gens])
    doc["gauss"] = np.asarray(MT.gaussian_weights(1.5, 5), dtype=np.float64)
    bump = h.copy()
    bump.points[:, 2] = 0.1 * (1 - bump.points[:, 0] ** 2 - bump.points[:, 1] ** 2)
    doc["flat_in"] = bump.points.copy()
    doc["flat_initial"] = np.asarray(MT.get_flat_initial(bump, axis=2))
    doc["flat_points_after"] = bump.points.copy()

    for sym in (True, False):
        tag = "sym" if sym else "full"
        # (end points as `Point`s, see above)
        g = B.ParametricCylindricalGuide(
            Point((0.0, 0.0, 0.0)), Point((0.0, 0.0, 5.0)), 0.5, theta_res=12, z_res=6, start_cap=False,
            end_cap=False, rotationally_symmetric=sym, initial_taper=(0.05, 0.25))
        # (no material_dict: the reference forwards its kwargs to cylindrical_mesh as well, which
        # rejects it, boundaries.py:1525-1534)
        if not sym:
            k = torch.arange(g.parameters.shape[0], dtype=torch.float64)
            g.parameters.assign_add(0.01 * torch.sin(1.7 * k))
        g.update()
        fv = torch.stack([g[k] for k in TRI], 1)
        if "guide_w" not in doc:
            doc["guide_w"] = rng.normal(size=tuple(fv.shape))
            doc["guide_wn"] = rng.normal(size=tuple(g["norm"].shape))
        loss = (fv * torch.tensor(doc["guide_w"])).sum() + (g["norm"] * torch.tensor(doc["guide_wn"])).sum()
        (grad,) = torch.autograd.grad(loss, [g.parameters])
        doc[f"guide_{tag}_params"] = g.parameters.numpy()
        doc[f"guide_{tag}_fields"], doc[f"guide_{tag}_norm"] = fv.detach().numpy(), g["norm"].detach().numpy()
        doc[f"guide_{tag}_grad"] = grad.numpy()
        doc[f"guide_{tag}_accumulator"] = np.asarray(g.accumulator, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "reference_mesh.npz"), **doc)
    print({k: np.shape(v) for k, v in doc.items()})


if __name__ == "__main__":
    main()
