#!/usr/bin/env python3
"""
Counterpart of the reference's dev/3d_trace.py: a dense point source (25 directions on a pi/8
cap x 6 wavelengths) shines on a short acrylic pyramid (read from an STL file, rotated after
loading) and an acrylic ball, in front of a square target.  ``ray_trace(6)`` with dead rays
compiled.

    python examples/trace_3d.py [--stl-dir DIR]

The reference loads ``./stl/short_pyramid.stl``; that file is not distributed with it, so this
script writes an equivalent square pyramid to ``--stl-dir`` first (binary STL) and reads it back
through ``ManualTriangleBoundary(file_name=...)``.
"""
import argparse
import os
import sys
import tempfile
from math import pi as PI

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tfrt.boundaries as boundaries          # noqa: E402
import tfrt.distributions as distributions    # noqa: E402
import tfrt.drawing as drawing                # noqa: E402
import tfrt.engine as engine                  # noqa: E402
import tfrt.materials as materials            # noqa: E402
import tfrt.mesh_tools as mt                  # noqa: E402
import tfrt.operation as operation            # noqa: E402
import tfrt.sources as sources                # noqa: E402


def short_pyramid(base=0.8, height=0.3):
    """Square pyramid: base in the x-y plane centred on the origin, apex on +z; closed,
    outward-facing triangles."""
    h = base / 2
    pts = np.array([[-h, -h, 0], [h, -h, 0], [h, h, 0], [-h, h, 0], [0, 0, height]], dtype=float)
    tri = [(0, 1, 4), (1, 2, 4), (2, 3, 4), (3, 0, 4), (0, 2, 1), (0, 3, 2)]
    cells = np.concatenate([np.full((6, 1), 3, dtype=np.int64), np.array(tri, dtype=np.int64)], 1)
    return mt.PolyData(pts, cells.reshape(-1))


def build(stl_dir=None, sphere_resolution=12, ray_dtype=None):
    angles = distributions.StaticUniformSphere(PI / 8.0, 25)
    source = sources.PointSource(3, (-1, 0, 0), (1, 0, 0), angles, drawing.RAINBOW_6, dense=True)
    source.frozen = True

    stl_dir = stl_dir or tempfile.mkdtemp(prefix="tfrt_stl_")
    os.makedirs(stl_dir, exist_ok=True)
    path = os.path.join(stl_dir, "short_pyramid.stl")
    short_pyramid().save(path)
    surface1 = boundaries.ManualTriangleBoundary(
        file_name=path, material_dict={"mat_in": 1, "mat_out": 0})
    surface2 = boundaries.ManualTriangleBoundary(
        mesh=mt.sphere(radius=.5, center=(1, 0, 0), theta_resolution=sphere_resolution,
                       phi_resolution=sphere_resolution),
        material_dict={"mat_in": 1, "mat_out": 0})
    surface3 = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(3, 0, 0), direction=(1, 0, 0), i_size=7, j_size=7).triangulate())

    system = engine.OpticalSystem3D()
    system.optical = [surface1, surface2]
    system.targets = [surface3]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()

    # as the reference script: turn the pyramid so its apex points at the source
    surface1.mesh.rotate_y(-90)
    surface1.update_from_mesh()
    system.update()

    trace_engine = engine.OpticalEngine(3, [operation.StandardReaction()], compile_dead_rays=True,
                                        dead_ray_length=10,
                                        **({} if ray_dtype is None else {"ray_dtype": ray_dtype}))
    trace_engine.optical_system = system
    trace_engine.validate_system()
    return trace_engine, system, (surface1, surface2, surface3), source


def main(stl_dir=None, max_iterations=6, verbose=True, ray_dtype=None):
    trace_engine, system, surfaces, source = build(stl_dir, ray_dtype=ray_dtype)
    trace_engine.ray_trace(max_iterations)
    if verbose:
        for name in ("active_rays", "finished_rays", "dead_rays", "unfinished_rays"):
            rays = getattr(trace_engine, name)
            print(f"{name:16s} {rays['x_start'].shape[0] if 'x_start' in rays.keys() else 0}")
    return trace_engine, system


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stl-dir", default=None)
    main(ap.parse_args().stl_dir)
