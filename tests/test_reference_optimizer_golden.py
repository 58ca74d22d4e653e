"""
Six optimiser steps against tests/golden/reference_optimizer.npz -- the same run made by the
reference's OWN stack (sources, distributions, boundaries + constraints, engine, operation,
materials, geometry, optimizer: SGD_Optimizer.single_step + smooth) executed in the build container
under tests/tf_shim (tests/golden/make_reference_optimizer_golden.py).

The product builds the scene through its tfrt-style API and must reproduce the mean error and both
parameter vectors after every step (float64 ray state, 1e-9), on the generic path (error function
as torch code through autograd), on the fused launch sequence and on its HIP-graph replay.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_optimizer.npz")


def _build(g, mode):
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.optimizer as optimizer
    import tfrt.sources as sources
    n = int(g["n_rays"])
    a = distributions.StaticUniformCircle(n, 0.2)
    distributions.BasePointTransformation(a, translation=(-10, 0, 0))
    b = distributions.StaticUniformCircle(n, 0.8)
    distributions.BasePointTransformation(b)
    source = sources.AperatureSource(3, a, b, [575.0], dense=False,
                                     extra_fields={"object_coords": ("start_point", a, "points")})
    lens = boundaries.ParametricMultiTriangleBoundary(
        mt.PolyData(g["points"], g["faces4"]), boundaries.FromVectorVG((1.0, 0.0, 0.0)),
        [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.2, "min")],
        [True, False], initial_parameters=[g["init0"], g["init1"]],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2, vertex_update_map=g["vmap"])
    target = boundaries.ManualTriangleBoundary(mesh=mt.PolyData(g["target_points"], g["target_faces4"]))
    system = engine.OpticalSystem3D()
    system.optical = lens.surfaces
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(3, [operation.StandardReaction()], ray_dtype=torch.float64,
                               simple_ray_inheritance={"wavelength", "object_coords"})
    eng.optical_system = system
    eng.validate_system()

    def torch_error(e):
        fin = e.finished_rays
        output = torch.stack([fin["y_end"], fin["z_end"]], dim=1)
        return (output - fin["object_coords"][:, 1:] * -1.0) ** 2

    erf = torch_error if mode == "generic" else optimizer.GoalError(
        ("y_end", "z_end"), lambda src: src["object_coords"][:, 1:] * -1.0)
    opt = optimizer.SGD_Optimizer(eng, lens.parameters, erf, 3, learning_rate=2e-4, grad_clip=0.05,
                                  fused=False if mode == "generic" else "auto",
                                  graph="auto" if mode == "graph" else False, speculative=False)
    return opt, lens, eng


@pytest.mark.parametrize("mode", ["generic", "fused", "graph"])
def test_six_optimiser_steps_reproduce_the_reference_stack(mode):
    g = np.load(GOLD)
    opt, lens, eng = _build(g, mode)
    if mode == "graph":
        opt._fused_step_warmup = None
    acc = [torch.as_tensor(g["accumulator"]), None]
    smoother = torch.as_tensor(g["smoother"])
    for step, lr in enumerate(g["lr"]):
        if mode == "graph" and opt._fused_step is not None:
            opt._fused_step.graph_warmup = 1            # capture early: most steps are replays
        err = float(opt.single_step(acc, lr_scale=float(lr)))
        if step % 2 == 1:
            opt.smooth(lens.parameters[0], smoother)
        assert abs(err - g["errors"][step]) <= 1e-9 * g["errors"][step], (step, err)
        for k, want in ((0, g["p0"][step]), (1, g["p1"][step])):
            got = lens.parameters[k].detach().cpu().numpy()
            assert np.abs(got - want).max() <= 1e-9, (step, k, np.abs(got - want).max())
    assert eng.finished_rays["x_start"].shape[0] == int(g["n_finished"])
    if mode == "graph":
        assert opt._fused_step.capture_error is None and opt._fused_step.graph_replays >= 2
