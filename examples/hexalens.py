#!/usr/bin/env python3
"""
Counterpart of the reference's dev/hexalens.py (most current optimisation script): a
two-surface parametric acrylic lens is shaped by gradient descent so that an object disc at
x = -10 is imaged (magnification -1) onto a target plane at x = +10.

Same scene topology as the reference script -- AperatureSource from two circle
distributions, ParametricMultiTriangleBoundary with two ThicknessConstraints and
flip_norm [True, False], vertex_update_map + gradient accumulator from
mesh_parametrization_tools, smoother from mesh_smoothing_tool, a phase schedule -- but the
per-step work (trace, error gradient) runs in the HIP kernels and the loop is driven by
SGD_Optimizer.training_routine instead of a hand-written tf.GradientTape loop.  No GUI.

    python examples/hexalens.py [--rays 20000] [--steps 30] [--edge 0.12]
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tfrt.boundaries as boundaries          # noqa: E402
import tfrt.distributions as distributions    # noqa: E402
import tfrt.drawing as drawing                # noqa: E402
import tfrt.engine as engine                  # noqa: E402
import tfrt.materials as materials            # noqa: E402
import tfrt.mesh_tools as mt                  # noqa: E402
import tfrt.operation as operation            # noqa: E402
import tfrt.optimizer as optimizer            # noqa: E402
import tfrt.sources as sources                # noqa: E402


def build(ray_count=20000, lens_res_scale=0.12, source_distance=10.0, magnification=1.0,
          object_size=0.2, lens_aperature=1.0, random_rays=True, generic_step=False):
    circle = distributions.RandomUniformCircle if random_rays else distributions.StaticUniformCircle
    start_points = circle(ray_count, object_size)
    distributions.BasePointTransformation(start_points, translation=(-source_distance, 0, 0))
    end_points = circle(ray_count, 0.98 * lens_aperature)
    distributions.BasePointTransformation(end_points)
    source = sources.AperatureSource(
        3, start_points, end_points, [drawing.YELLOW], dense=False,
        extra_fields={"object_coords": ("start_point", start_points, "points")})

    zero_points = mt.circular_mesh(lens_aperature, lens_res_scale)
    zero_points.rotate_y(90)
    zero_points.rotate_x(90)
    top_parent = mt.get_closest_point(zero_points, (0, 0, 0))
    vertex_update_map, accumulator = mt.mesh_parametrization_tools(zero_points, top_parent)

    lens = boundaries.ParametricMultiTriangleBoundary(
        zero_points, boundaries.FromVectorVG((1, 0, 0)),
        [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.2, "min")],
        [True, False],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2,
        vertex_update_map=vertex_update_map)
    target = boundaries.ManualTriangleBoundary(mesh=mt.plane(
        center=(source_distance * magnification, 0, 0), direction=(1, 0, 0), i_size=100, j_size=100))
    target.frozen = True

    system = engine.OpticalSystem3D()
    system.optical = lens.surfaces
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()

    trace_engine = engine.OpticalEngine(
        3, [operation.StandardReaction()], compile_active_rays=False,
        simple_ray_inheritance={"wavelength", "object_coords"})
    trace_engine.optical_system = system
    trace_engine.validate_system()

    def torch_error_function(eng):
        """dev/hexalens.py:144-168 (inner goal) as the reference writes it: arbitrary code on the
        finished rays (the generic step: reads the ray counts back, autograd through torch ops)."""
        fin = eng.finished_rays
        output = torch.stack([fin["y_end"], fin["z_end"]], dim=1).double()
        goal = fin["object_coords"][:, 1:] * -magnification
        return (output - goal) ** 2

    # the same error stated as a GoalError: squared_difference(stack(finished[fields]), goal) with the
    # goal a function of inherited source fields -- the optimiser then runs the step as one fixed
    # launch sequence replayed from a HIP graph (tensorflowraytrace_amd/fused_step.py)
    error_function = torch_error_function if generic_step else optimizer.GoalError(
        ("y_end", "z_end"), lambda src: src["object_coords"][:, 1:] * -magnification)

    smoother = mt.mesh_smoothing_tool(zero_points, [300, 50, 20, 10, 5])
    return dict(engine=trace_engine, system=system, lens=lens, error_function=error_function,
                accumulator=accumulator, smoother=smoother, zero_points=zero_points)


def save_parameters(lens, filename, history=None):
    """Parameter-history file of dev/hexalens.py:305-306, 338-347: a pickled list of
    ``(first_surface_parameters, second_surface_parameters)`` numpy tuples, one per record."""
    if history is None:
        history = [tuple(p.detach().cpu().numpy() for p in lens.parameters)]
    with open(filename, "wb") as f:
        pickle.dump(history, f, pickle.HIGHEST_PROTOCOL)


def load_parameters(lens, system, filename):
    """dev/hexalens.py:327-336: restore both surfaces from the last record of a history file."""
    with open(filename, "rb") as f:
        history = pickle.load(f)
    with torch.no_grad():
        for p, saved in zip(lens.parameters, history[-1]):
            p.copy_(torch.as_tensor(np.asarray(saved), dtype=p.dtype, device=p.device))
    system.update()
    return history


def save_meshes(lens, directory):
    """dev/hexalens.py:322-325: both lens surfaces as STL files."""
    os.makedirs(directory, exist_ok=True)
    for surface in lens.surfaces:       # the reference's drawers do this before every redraw
        surface.update_mesh_from_vertices()
    lens.surfaces[0].save(os.path.join(directory, "hexalens_first.stl"))
    lens.surfaces[1].save(os.path.join(directory, "hexalens_second.stl"))


def run(ray_count=20000, steps=30, lens_res_scale=0.12, verbose=True, history_file=None,
        resume_from=None, generic_step=False):
    s = build(ray_count, lens_res_scale, generic_step=generic_step)
    parameter_history = []
    if resume_from:
        parameter_history = load_parameters(s["lens"], s["system"], resume_from)
    opt = optimizer.SGD_Optimizer(s["engine"], s["lens"].parameters, s["error_function"], 3,
                                  learning_rate=2e-5 * (20000 / ray_count), grad_clip=1.0)
    opt.suppress_warnings = True
    errors = []

    def record():
        errors.append(opt.last_error)
        if history_file and opt.iterations % 10 == 0:       # dev/hexalens.py:294-303
            parameter_history.append(tuple(p.detach().cpu().numpy() for p in s["lens"].parameters))

    # wrap single_step to keep the per-step mean error
    orig = opt.single_step

    def single_step(*a, **k):
        opt.last_error = orig(*a, **k)
        return opt.last_error

    opt.single_step = single_step
    routine = [
        {"steps": max(steps // 2, 1), "learning_rate": 1.0, "accumulators": s["accumulator"],
         "smoothers": s["smoother"]},
        {"steps": max(steps - steps // 2, 1), "learning_rate": (1.0, 0.5), "accumulators": None,
         "smoothers": None},
    ]
    opt.training_routine(routine, post_step=record, report_frequency=5 if verbose else 0,
                         show_time=verbose)
    if history_file:
        parameter_history.append(tuple(p.detach().cpu().numpy() for p in s["lens"].parameters))
        save_parameters(s["lens"], history_file, parameter_history)
    return errors, s


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--edge", type=float, default=0.12)
    ap.add_argument("--history", default=None, help="write the parameter history (pickle) here")
    ap.add_argument("--resume", default=None, help="start from the last record of this history")
    ap.add_argument("--stl-dir", default=None, help="write both optimised surfaces as STL here")
    ap.add_argument("--generic-step", action="store_true",
                    help="error function as torch code (the reference's form) instead of a GoalError")
    a = ap.parse_args()
    errs, state = run(a.rays, a.steps, a.edge, history_file=a.history, resume_from=a.resume,
                      generic_step=a.generic_step)
    if a.stl_dir:
        save_meshes(state["lens"], a.stl_dir)
    print(f"mean squared image error: first {errs[0]:.6g} -> last {errs[-1]:.6g}")
