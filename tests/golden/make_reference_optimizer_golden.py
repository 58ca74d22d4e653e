#!/usr/bin/env python3
"""
An optimisation run of the reference's own stack, executed under tests/tf_shim: reference
AperatureSource (3-D, undense, inherited ``object_coords``) -> reference
ParametricMultiTriangleBoundary (two hex-mesh surfaces, ThicknessConstraints, vertex_update_map)
+ ManualTriangleBoundary target -> reference OpticalSystem3D / OpticalEngine / StandardReaction ->
the error function of dev/hexalens.py:144-168 (inner goal) -> reference SGD_Optimizer.single_step
(tf.GradientTape, non-finite -> 0, scale, clip, accumulator matmul, Keras SGD apply) and
SGD_Optimizer.smooth, for six steps with a learning-rate schedule.  Writes
tests/golden/reference_optimizer.npz: the inputs, and per step the mean error and both parameter
vectors.

What the stand-in supplies besides arithmetic: tf.GradientTape (torch.autograd) and the Keras SGD
update ``var -= 0.01 * grad`` (momentum assigned after construction is inert in Keras OptimizerV2;
see tests/tf_shim).  The accumulator and smoother matrices are inputs (built with this package's
mesh tools: the reference's need pyvista).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, os.path.join(TESTS, "tf_shim"))
if not os.path.exists("/root/reference/tfrt/optimizer.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import pyvista as pv                     # noqa: E402  (placeholder)
import tensorflow as tf                  # noqa: E402  (stand-in)
import tfrt.boundaries as B              # noqa: E402  (the reference's modules)
import tfrt.distributions as D           # noqa: E402
import tfrt.engine as E                  # noqa: E402
import tfrt.materials as M               # noqa: E402
import tfrt.operation as O               # noqa: E402
import tfrt.optimizer as OPT             # noqa: E402
import tfrt.sources as S                 # noqa: E402
import scene_util                        # noqa: E402


def inputs():
    """Everything both sides are given (numpy)."""
    pts, faces = scene_util.hex_mesh(3)
    P = np.stack([np.zeros(len(pts)), pts[:, 0], pts[:, 1]], 1)
    f4 = np.concatenate([np.full((len(faces), 1), 3), faces], 1).reshape(-1)
    r2 = P[:, 1] ** 2 + P[:, 2] ** 2
    rng = np.random.default_rng(5)
    vmap = rng.uniform(size=(len(faces), 3)) > 0.2
    # accumulator / smoother: any square matrices do; these are the mesh tools' (ancestor sums,
    # neighbour smoothing), built with this package because the reference's need pyvista
    sys.path.insert(0, ROOT)
    import tensorflowraytrace_amd.mesh_tools as mt
    mesh = mt.PolyData(P, f4)
    _, acc = mt.mesh_parametrization_tools(mesh, 0)
    smoother = mt.mesh_smoothing_tool(mesh, [8, 2, 1])
    sys.path.remove(ROOT)
    t = 10.0
    target_points = np.array([[t, -50.0, -50.0], [t, 50.0, -50.0], [t, 50.0, 50.0], [t, -50.0, 50.0]])
    target_faces = np.array([3, 0, 1, 2, 3, 0, 2, 3])
    return dict(points=P, faces4=f4, vmap=vmap, init0=-0.15 * (1 - r2), init1=0.15 * (1 - r2),
                accumulator=np.asarray(acc, dtype=np.float64), smoother=np.asarray(smoother, dtype=np.float64),
                target_points=target_points, target_faces4=target_faces,
                n_rays=np.int64(900), lr=np.array([1.0, 0.9, 0.8, 0.7, 0.6, 0.5]))


def main():
    d = inputs()
    n = int(d["n_rays"])
    a = D.StaticUniformCircle(n, 0.2)
    D.BasePointTransformation(a, translation=(-10, 0, 0))
    b = D.StaticUniformCircle(n, 0.8)
    D.BasePointTransformation(b)
    source = S.AperatureSource(3, a, b, [575.0], dense=False,
                               extra_fields={"object_coords": ("start_point", a, "points")})
    lens = B.ParametricMultiTriangleBoundary(
        pv.PolyData(d["points"], d["faces4"]), B.FromVectorVG((1.0, 0.0, 0.0)),
        [B.ThicknessConstraint(0.0, "min"), B.ThicknessConstraint(0.2, "min")], [True, False],
        initial_parameters=[d["init0"], d["init1"]], material_list=[{"mat_in": 1, "mat_out": 0}] * 2,
        vertex_update_map=d["vmap"])
    target = B.ManualTriangleBoundary(mesh=pv.PolyData(d["target_points"], d["target_faces4"]))
    system = E.OpticalSystem3D()
    system.optical = lens.surfaces
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": M.vacuum}, {"n": M.acrylic}]
    system.update()
    eng = E.OpticalEngine(3, [O.StandardReaction()],
                          simple_ray_inheritance={"wavelength", "object_coords"})
    eng.optical_system = system
    eng.validate_system()

    def error_function(engine):
        fin = engine.finished_rays
        output = tf.stack([fin["y_end"], fin["z_end"]], axis=1)
        goal = fin["object_coords"][:, 1:] * -1.0
        return (output - goal) ** 2

    opt = OPT.SGD_Optimizer(eng, lens.parameters, error_function, 3, learning_rate=2e-4, grad_clip=0.05)
    acc = [tf.constant(d["accumulator"]), None]
    errors, p0s, p1s = [], [], []
    for step, lr in enumerate(d["lr"]):
        err = opt.single_step(acc, lr_scale=float(lr))
        if step % 2 == 1:
            opt.smooth(lens.parameters[0], tf.constant(d["smoother"]))
        errors.append(float(err))
        p0s.append(lens.parameters[0].numpy().copy())
        p1s.append(lens.parameters[1].numpy().copy())
    d.update(errors=np.array(errors), p0=np.stack(p0s), p1=np.stack(p1s),
             n_finished=np.int64(eng.finished_rays["x_start"].shape[0]))
    np.savez_compressed(os.path.join(HERE, "reference_optimizer.npz"), **d)
    print("errors", errors, "finished", int(d["n_finished"]))


if __name__ == "__main__":
    main()
