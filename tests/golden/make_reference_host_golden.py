#!/usr/bin/env python3
"""
Golden vectors for the host-side classes either side of the hot path, produced by EXECUTING the
reference's own tfrt/boundaries.py, tfrt/sources.py and tfrt/distributions.py under tests/tf_shim
(+ placeholder pyvista / imageio modules: both are only imported, never used here).  Writes
tests/golden/reference_host.npz:

* boundaries.py: ParametricTriangleBoundary (flip_norm, vertex_update_map; fields + the gradient
  of a fixed linear functional of the fields w.r.t. the parameters), ParametricMultiTriangleBoundary
  with ThicknessConstraints, MasterSlaveParametricTriangleBoundary, ParametricSegmentBoundary /
  ParametricMultiSegmentBoundary (2-D), PointConstraint / ThicknessConstraint / ClipConstraint on
  bare parameter holders, SecondSurfaceVG / FromPointVG / FromVectorVG / FromAxisVG
* sources.py + distributions.py: StaticUniformCircle / Beam / AngularDistribution points, 3-D
  AperatureSource undense (with an inherited extra field) and dense (4 x 5 x 2)

The reference's 2-D PointSource / AngularSource cannot be constructed at HEAD (their ``center``
setter rejects every 2-vector, sources.py:447-449, 660-662), so they have no vectors here.
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
if not os.path.exists("/root/reference/tfrt/boundaries.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import pyvista as pv                    # noqa: E402  (the placeholder of tests/tf_shim)
import tensorflow as tf                 # noqa: E402  (the stand-in)
import tfrt.boundaries as B             # noqa: E402  (the reference's modules)
import tfrt.distributions as D          # noqa: E402
import tfrt.sources as S                # noqa: E402
import scene_util                       # noqa: E402

TRI = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2")
SEG = ("x_start", "y_start", "x_end", "y_end")


def hex_mesh(k):
    pts, faces = scene_util.hex_mesh(k)
    P = np.stack([np.zeros(len(pts)), pts[:, 0], pts[:, 1]], 1)
    f4 = np.concatenate([np.full((len(faces), 1), 3), faces], 1).reshape(-1)
    return P, faces, f4


def tri_fields(b):
    return torch.stack([b[k] for k in TRI], 1).detach().numpy(), b["norm"].detach().numpy()


def functional(b, w9, w3):
    fv = torch.stack([b[k] for k in TRI], 1)
    return (fv * torch.tensor(w9)).sum() + (b["norm"] * torch.tensor(w3)).sum()


def main():
    rng = np.random.default_rng(77)
    doc = {}

    # ---------------------------------------------------------------- triangle boundaries
    P, faces, f4 = hex_mesh(3)
    r2 = P[:, 1] ** 2 + P[:, 2] ** 2
    vmap = rng.uniform(size=(len(faces), 3)) > 0.3
    init = -(0.1 + 0.15 * (1 - r2)) + 0.01 * np.sin(7 * np.arange(len(P)))
    w9, w3 = rng.normal(size=(len(faces), 9)), rng.normal(size=(len(faces), 3))
    doc.update(hex_points=P, hex_faces=faces, ptb_vmap=vmap, ptb_init=init, ptb_w9=w9, ptb_w3=w3)
    b = B.ParametricTriangleBoundary(pv.PolyData(P, f4), B.FromVectorVG((1.0, 0.0, 0.0)), flip_norm=True,
                                     initial_parameters=init, vertex_update_map=vmap,
                                     material_dict={"mat_in": 1, "mat_out": 0})
    b.update()
    doc["ptb_fields"], doc["ptb_norm"] = tri_fields(b)
    (g,) = torch.autograd.grad(functional(b, w9, w3), [b.parameters])
    doc["ptb_grad"] = g.numpy()
    doc["ptb_mat_in"] = b["mat_in"].numpy()

    # multi surface with thickness constraints (boundaries.py:1233-1412, 162-215)
    init0, init1 = -0.15 * (1 - r2) + 0.03, 0.15 * (1 - r2) - 0.05
    m = B.ParametricMultiTriangleBoundary(
        pv.PolyData(P, f4), B.FromVectorVG((1.0, 0.0, 0.0)),
        [B.ThicknessConstraint(0.0, "min"), B.ThicknessConstraint(0.2, "min")], [True, False],
        initial_parameters=[init0, init1], material_list=[{"mat_in": 1, "mat_out": 0}] * 2)
    m.update()
    doc.update(multi_init0=init0, multi_init1=init1,
               multi_p0=m.surfaces[0].parameters.numpy(), multi_p1=m.surfaces[1].parameters.numpy())
    doc["multi_fields0"], doc["multi_norm0"] = tri_fields(m.surfaces[0])
    doc["multi_fields1"], doc["multi_norm1"] = tri_fields(m.surfaces[1])

    # master / slave (boundaries.py:1116-1229): mirror symmetry in y
    def filter_masters(verts):
        v = np.asarray(verts)
        return [int(i) for i in np.nonzero(v[:, 1] >= -1e-9)[0]]

    def attach_slaves(verts, master, available):
        v = np.asarray(verts)
        mm = v[master]
        return {s for s in available if abs(v[s, 1] + mm[1]) < 1e-9 and abs(v[s, 2] - mm[2]) < 1e-9}

    ms = B.MasterSlaveParametricTriangleBoundary(
        filter_masters, attach_slaves, pv.PolyData(P, f4), B.FromVectorVG((1.0, 0.0, 0.0)),
        flip_norm=False, initial_parameters=init, material_dict={"mat_in": 1, "mat_out": 0})
    ms.update()
    doc["ms_params"] = ms.parameters.numpy()
    doc["ms_gather"] = ms._gather.numpy()
    doc["ms_fields"], doc["ms_norm"] = tri_fields(ms)
    (g,) = torch.autograd.grad(functional(ms, w9, w3), [ms.parameters])
    doc["ms_grad"] = g.numpy()

    # ---------------------------------------------------------------- constraints on bare holders
    class Holder:
        def __init__(self, p):
            self.parameters = tf.Variable(np.asarray(p, dtype=np.float64))

    pa, pb = rng.normal(size=9), rng.normal(size=9)
    doc.update(con_a=pa, con_b=pb)
    for tag, con in (("thick_min", B.ThicknessConstraint(0.25, "min")),
                     ("thick_max", B.ThicknessConstraint(0.25, "max")),
                     ("point", B.PointConstraint(0.3, 4)),
                     ("point_pv", B.PointConstraint(-0.1, 2, parent_vertex=6))):
        hs = [Holder(pa), Holder(pb)]
        con.make(1, hs)()
        doc["con_" + tag] = hs[1].parameters.numpy()
    h0 = [Holder(pa), Holder(pb)]
    B.ThicknessConstraint(0.1, "min").make(0, h0)()          # target 0 with parent "prev": against zero
    doc["con_first"] = h0[0].parameters.numpy()
    hz = Holder(pb)
    B.ThicknessConstraint(0.05, "max", parent="zero").make(hz, None)()
    doc["con_zero"] = hz.parameters.numpy()
    hc = Holder(pa)
    B.ClipConstraint(-0.5, 0.7).make(hc, None)()
    doc["con_clip"] = hc.parameters.numpy()

    # ---------------------------------------------------------------- vector generators
    zero = tf.constant(P + np.array([0.0, 0.0, 0.0]))
    second = P + np.array([1.0, 0.0, 0.0]) + 0.2 * P[:, [2, 1, 0]]
    doc["vg_second_points"] = second
    doc["vg_second"] = B.SecondSurfaceVG(second).generate(zero).numpy()
    doc["vg_point"] = B.FromPointVG((-4.0, 0.1, -0.05)).generate(zero).numpy()
    doc["vg_vector"] = B.FromVectorVG((0.3, -0.4, 1.2)).generate(zero).numpy()
    doc["vg_axis"] = B.FromAxisVG(tf.constant((-3.0, 0.0, 0.0)), direction=tf.constant((0.0, 0.0, 1.0))).generate(zero).numpy()

    # ---------------------------------------------------------------- 2-D parametric segments
    k = 13
    ys = np.linspace(-1.1, 1.1, k)
    zp, op = np.stack([np.zeros(k), ys], 1), np.stack([np.ones(k) + 0.1 * ys, ys * 1.05], 1)
    zd = D.ManualBasePointDistribution(2, points=tf.constant(zp))   # (tensors: numpy + stand-in tensor
    od = D.ManualBasePointDistribution(2, points=tf.constant(op))   # arithmetic is not defined)
    sp = -(0.1 + 0.2 * (1 - (ys / 1.1) ** 2))
    ws = rng.normal(size=(k - 1, 4))
    doc.update(seg_zero=zp, seg_one=op, seg_init=sp, seg_w=ws)
    # (ParametricSegmentBoundary cannot be instantiated at the reference's HEAD:
    # SegmentBoundaryBase.update_materials passes `self` twice, boundaries.py:487.  Its arithmetic is
    # the static _update_internal (boundaries.py:611-617), which is what runs here; the thickness
    # constraint of a two-layer multi boundary is applied with the constraint classes themselves.)
    for flip in (False, True):
        p = tf.Variable(sp)
        out = B.ParametricSegmentBoundary._update_internal(zd.points, od.points, p, flip)
        fields = torch.stack(list(out), 1)
        (g,) = torch.autograd.grad((fields * torch.tensor(ws)).sum(), [p])
        doc[f"seg_fields_{int(flip)}"], doc[f"seg_grad_{int(flip)}"] = fields.detach().numpy(), g.numpy()
    bump = 1 - (ys / 1.1) ** 2

    class Layer:
        def __init__(self, p):
            self.parameters = tf.Variable(p)

    layers = [Layer(-0.2 * bump - 0.05), Layer(0.2 * bump)]
    doc["mseg_init0"], doc["mseg_init1"] = -0.2 * bump - 0.05, 0.2 * bump
    B.ThicknessConstraint(0.0, "min").make(0, layers)()
    B.ThicknessConstraint(0.15, "min").make(1, layers)()
    doc["mseg_p0"], doc["mseg_p1"] = layers[0].parameters.numpy(), layers[1].parameters.numpy()
    parts = [torch.stack(list(B.ParametricSegmentBoundary._update_internal(zd.points, od.points, l.parameters, f)), 1)
             for l, f in zip(layers, (True, False))]
    doc["mseg_fields"] = torch.cat(parts, 0).detach().numpy()

    # ---------------------------------------------------------------- distributions and sources
    c = D.StaticUniformCircle(11, 0.2)
    doc["dist_circle"] = c.points.numpy()
    doc["dist_beam"] = D.StaticUniformBeam(-1.5, 1.5, 10).points.numpy()
    doc["dist_angles"] = D.StaticUniformAngularDistribution(-0.1, 0.25, 7).angles.numpy()
    a = D.StaticUniformCircle(11, 0.2)
    D.BasePointTransformation(a, translation=(-10, 0, 0))
    b2 = D.StaticUniformCircle(11, 0.9)
    D.BasePointTransformation(b2)
    src = S.AperatureSource(3, a, b2, [575.0], dense=False,
                            extra_fields={"object_coords": ("start_point", a, "points")})
    src.update()
    for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength", "object_coords"):
        doc["ap_undense_" + f] = src[f].numpy()
    a = D.StaticUniformCircle(4, 0.2)
    D.BasePointTransformation(a, translation=(-3, 0, 0))
    b2 = D.StaticUniformCircle(5, 0.9)
    D.BasePointTransformation(b2)
    tag = np.arange(5, dtype=np.float64) * 10
    src = S.AperatureSource(3, a, b2, [450.0, 650.0], dense=True,
                            extra_fields={"end_tag": ("end_point", tag)})
    src.update()
    for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength", "end_tag"):
        doc["ap_dense_" + f] = src[f].numpy()

    np.savez_compressed(os.path.join(HERE, "reference_host.npz"), **doc)
    print(len(doc), "arrays;", sorted(doc)[:8], "...")


if __name__ == "__main__":
    main()
