#!/usr/bin/env python3
"""
Counterpart of the reference's dev/single_pass.py (BASELINE config 1): a 2-D beam of 10 rays
x 6 wavelengths hits one acrylic arc; one ``single_pass`` is run and the structure of the
projection result is printed.  (The reference script also draws the result; no GUI here.)
"""
import os
import sys
from math import pi as PI

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tfrt.boundaries as boundaries          # noqa: E402
import tfrt.distributions as distributions    # noqa: E402
import tfrt.drawing as drawing                # noqa: E402
import tfrt.engine as eng                     # noqa: E402
import tfrt.materials as materials            # noqa: E402
import tfrt.operation as op                   # noqa: E402
import tfrt.sources as sources                # noqa: E402


def main():
    arc_boundary = boundaries.ManualArcBoundary()
    arc_boundary["x_center"] = np.array([5], dtype=np.float64)
    arc_boundary["y_center"] = np.array([0], dtype=np.float64)
    arc_boundary["angle_start"] = np.array([3 * PI / 4], dtype=np.float64)
    arc_boundary["angle_end"] = np.array([5 * PI / 4], dtype=np.float64)
    arc_boundary["radius"] = np.array([5], dtype=np.float64)
    eng.annotation_helper(arc_boundary, "mat_in", 1, "x_center", dtype=torch.int64)
    eng.annotation_helper(arc_boundary, "mat_out", 0, "x_center", dtype=torch.int64)

    beam_points = distributions.StaticUniformBeam(-1.5, 1.5, 10)
    angles = distributions.StaticUniformAngularDistribution(0, 0, 1)
    source = sources.AngularSource(2, (-1.0, 0.0), 0.0, angles, beam_points, drawing.RAINBOW_6)

    system = eng.OpticalSystem2D()
    system.optical_arcs = [arc_boundary]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]

    trace_engine = eng.OpticalEngine(2, [op.StandardReaction()], compile_dead_rays=True,
                                     dead_ray_length=10, simple_ray_inheritance={"wavelength"})
    trace_engine.optical_system = system
    system.update()
    trace_engine.validate_system()

    new_rays = trace_engine.single_pass(dict(system._amalgamated_sources))
    print("projected result printout")
    eng.recursive_dict_key_print(trace_engine.last_projection_result)
    print("------------")
    print("new rays printout")
    eng.recursive_dict_key_print(new_rays)
    return trace_engine, new_rays


if __name__ == "__main__":
    main()
