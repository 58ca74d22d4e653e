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


def trace(sets, rays, wl, passes):
    system = ref_engine.OpticalSystem2D()
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
    for cls, rs in (("finished", eng.finished_rays), ("active", eng.active_rays),
                    ("stopped", eng.stopped_rays), ("dead", eng.dead_rays)):
        if bool(rs):
            out[cls] = torch.stack([rs[g] for g in GEO]).numpy()
            out[cls + "_id"] = rs["ray_id"].numpy().astype(np.int64)
        else:
            out[cls] = np.zeros((4, 0))
            out[cls + "_id"] = np.zeros(0, dtype=np.int64)
    return out


def main():
    doc = {}
    for tag, seed, with_seg, with_arc in (("arc", 11, False, True), ("seg", 12, True, False),
                                          ("both", 13, True, True)):
        rng = np.random.default_rng(seed)
        sets, rays, wl = _scene(rng, 1500, with_seg=with_seg, with_arc=with_arc)
        out = trace(sets, rays, wl, 4)
        doc[f"{tag}_rays"], doc[f"{tag}_wl"] = rays, wl
        for name, fields in sets.items():
            for f, v in fields.items():
                doc[f"{tag}__{name}__{f}"] = v.numpy()
        for k, v in out.items():
            doc[f"{tag}_{k}"] = v
        print(tag, {k: v.shape for k, v in out.items() if not k.endswith("_id")})
    np.savez_compressed(os.path.join(HERE, "reference_trace2d.npz"), **doc)


if __name__ == "__main__":
    main()
