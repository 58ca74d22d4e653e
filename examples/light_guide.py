#!/usr/bin/env python3
"""
Counterpart of the reference's dev/light_guide.py: a thin acrylic wedge (three segments) is fed
from its narrow base by a Lambertian fan of rays; the rays bounce down the wedge by total
internal reflection until they leak out.  ``ray_trace(50)`` with dead rays compiled.

    python examples/light_guide.py [--rays 100] [--plot guide.png]

The reference script opens a matplotlib window; here the ray sets are returned / summarised and
``--plot`` writes an image instead.
"""
import argparse
import os
import sys
from math import pi as PI

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tfrt.boundaries as boundaries          # noqa: E402
import tfrt.distributions as distributions    # noqa: E402
import tfrt.drawing as drawing                # noqa: E402
import tfrt.engine as eng                     # noqa: E402
import tfrt.materials as materials            # noqa: E402
import tfrt.operation as op                   # noqa: E402
import tfrt.sources as sources                # noqa: E402

WEDGE = [(-.1, -4, 0, 4), (0, 4, .1, -4), (.1, -4, -.1, -4)]


def build(sample_count=100, random=True, ray_dtype=None):
    boundary = boundaries.ManualSegmentBoundary()
    boundary.feed_segments(WEDGE)
    boundary["mat_in"] = np.array((1, 1, 1), dtype=np.int64)
    boundary["mat_out"] = np.array((0, 0, 0), dtype=np.int64)

    if random:
        angles = distributions.RandomLambertianAngularDistribution(-.4 * PI, .4 * PI, sample_count)
        beam_points = distributions.RandomUniformBeam(-.09, .09, sample_count)
    else:
        angles = distributions.StaticLambertianAngularDistribution(-.4 * PI, .4 * PI, sample_count)
        beam_points = distributions.StaticUniformBeam(-.09, .09, sample_count)
    source = sources.AngularSource(
        2, (0, -4.001), PI / 2, angles, beam_points, [drawing.YELLOW] * sample_count,
        rank_type=None, dense=False)

    system = eng.OpticalSystem2D()
    system.optical_segments = [boundary]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]

    trace_engine = eng.OpticalEngine(2, [op.StandardReaction()], compile_dead_rays=True,
                                     dead_ray_length=10, simple_ray_inheritance={"wavelength"},
                                     **({} if ray_dtype is None else {"ray_dtype": ray_dtype}))
    trace_engine.optical_system = system
    system.update()
    trace_engine.validate_system()
    return trace_engine, system, boundary, source


def main(sample_count=100, max_iterations=50, random=True, plot=None, verbose=True,
         ray_dtype=None):
    trace_engine, system, boundary, source = build(sample_count, random, ray_dtype)
    trace_engine.ray_trace(max_iterations=max_iterations)
    if verbose:
        for name in ("active_rays", "finished_rays", "dead_rays", "unfinished_rays"):
            rays = getattr(trace_engine, name)
            print(f"{name:16s} {rays['x_start'].shape[0] if 'x_start' in rays.keys() else 0}")
    if plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(1, 1, figsize=(9, 9))
        ax.set_aspect("equal")
        rays = trace_engine.all_rays
        xs = np.stack([rays["x_start"].cpu().numpy(), rays["x_end"].cpu().numpy()])
        ys = np.stack([rays["y_start"].cpu().numpy(), rays["y_end"].cpu().numpy()])
        ax.plot(xs, ys, color="gold", linewidth=0.4)
        for x0, y0, x1, y1 in WEDGE:
            ax.plot([x0, x1], [y0, y1], color="c")
        ax.set_xlim(-2, 12)
        ax.set_ylim(-7, 7)
        fig.savefig(plot, dpi=120)
    return trace_engine, system


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=100)
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--plot", default=None)
    a = ap.parse_args()
    main(a.rays, a.iterations, plot=a.plot)
