"""
The host-side classes either side of the hot path against tests/golden/reference_host.npz -- the
outputs of the reference's OWN tfrt/boundaries.py, tfrt/sources.py and tfrt/distributions.py,
executed in the build container under tests/tf_shim (tests/golden/make_reference_host_golden.py):

* ParametricTriangleBoundary (flip_norm, vertex_update_map): face fields, normals, and the gradient
  of a fixed linear functional of them w.r.t. the parameters (the stop_gradient mask at work);
* ParametricMultiTriangleBoundary + ThicknessConstraint; MasterSlaveParametricTriangleBoundary;
* PointConstraint / ThicknessConstraint (min, max, first surface, parent="zero") / ClipConstraint;
* SecondSurfaceVG / FromPointVG / FromVectorVG / FromAxisVG;
* the 2-D parametric segment arithmetic (boundaries.py:611-617) and its thickness constraints;
* StaticUniformCircle / Beam / AngularDistribution and the 3-D AperatureSource, undense with an
  inherited extra field and dense (ray ORDER included).

Runs on the CPU (face construction through the oracle-backed stand-ins of tests/cpu_backend.py:
this checks the product's host logic) and, marked gpu, with the HIP kernels underneath.
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_host.npz")
TRI = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2")
SEG = ("x_start", "y_start", "x_end", "y_end")


def _np(t):
    return t.detach().cpu().numpy()


def _close(got, want, tol=1e-14, msg=""):
    np.testing.assert_allclose(_np(got) if isinstance(got, torch.Tensor) else got, want, rtol=0,
                               atol=tol * max(1.0, float(np.abs(want).max())), err_msg=msg)


def _check_all():
    import tensorflowraytrace_amd.boundaries as B
    import tensorflowraytrace_amd.distributions as D
    import tensorflowraytrace_amd.mesh_tools as mt
    import tensorflowraytrace_amd.sources as S
    from tensorflowraytrace_amd import config
    g = np.load(GOLD)
    dev = config.get_device()
    P, faces = g["hex_points"], g["hex_faces"]
    f4 = np.concatenate([np.full((len(faces), 1), 3), faces], 1).reshape(-1)
    mesh = lambda: mt.PolyData(P, f4)

    def fields(b):
        return torch.stack([b[k] for k in TRI], 1)

    def functional(b):
        return (fields(b) * torch.tensor(g["ptb_w9"], device=dev)).sum() + \
            (b["norm"] * torch.tensor(g["ptb_w3"], device=dev)).sum()

    # ---- ParametricTriangleBoundary
    b = B.ParametricTriangleBoundary(mesh(), B.FromVectorVG((1.0, 0.0, 0.0)), flip_norm=True,
                                     initial_parameters=g["ptb_init"], vertex_update_map=g["ptb_vmap"],
                                     material_dict={"mat_in": 1, "mat_out": 0})
    b.update()
    _close(fields(b), g["ptb_fields"], msg="ptb fields")
    _close(b["norm"], g["ptb_norm"], msg="ptb norm")
    assert np.array_equal(_np(b["mat_in"]), g["ptb_mat_in"])
    (grad,) = torch.autograd.grad(functional(b), [b.parameters])
    _close(grad, g["ptb_grad"], 1e-12, "ptb gradient (vertex_update_map)")

    # ---- multi surface + thickness constraints
    m = B.ParametricMultiTriangleBoundary(
        mesh(), B.FromVectorVG((1.0, 0.0, 0.0)),
        [B.ThicknessConstraint(0.0, "min"), B.ThicknessConstraint(0.2, "min")], [True, False],
        initial_parameters=[g["multi_init0"], g["multi_init1"]],
        material_list=[{"mat_in": 1, "mat_out": 0}] * 2)
    m.update()
    _close(m.surfaces[0].parameters, g["multi_p0"], msg="multi p0")
    _close(m.surfaces[1].parameters, g["multi_p1"], msg="multi p1")
    for k, s in enumerate(m.surfaces):
        _close(fields(s), g[f"multi_fields{k}"], msg=f"multi fields {k}")
        _close(s["norm"], g[f"multi_norm{k}"], msg=f"multi norm {k}")

    # ---- master / slave
    def filter_masters(verts):
        v = _np(verts) if isinstance(verts, torch.Tensor) else np.asarray(verts)
        return [int(i) for i in np.nonzero(v[:, 1] >= -1e-9)[0]]

    def attach_slaves(verts, master, available):
        v = _np(verts) if isinstance(verts, torch.Tensor) else np.asarray(verts)
        mm = v[master]
        return {s for s in available if abs(v[s, 1] + mm[1]) < 1e-9 and abs(v[s, 2] - mm[2]) < 1e-9}

    ms = B.MasterSlaveParametricTriangleBoundary(
        filter_masters, attach_slaves, mesh(), B.FromVectorVG((1.0, 0.0, 0.0)), flip_norm=False,
        initial_parameters=g["ptb_init"], material_dict={"mat_in": 1, "mat_out": 0})
    ms.update()
    _close(ms.parameters, g["ms_params"], msg="master parameters")
    assert np.array_equal(_np(ms._gather), g["ms_gather"])
    _close(fields(ms), g["ms_fields"], msg="ms fields")
    (grad,) = torch.autograd.grad(functional(ms), [ms.parameters])
    _close(grad, g["ms_grad"], 1e-12, "master/slave gradient (reverse of the gather)")

    # ---- constraints on bare parameter holders
    from tensorflowraytrace_amd.variable import Variable

    class Holder:
        def __init__(self, p):
            self.parameters = Variable(np.asarray(p))

    for tag, con in (("thick_min", B.ThicknessConstraint(0.25, "min")),
                     ("thick_max", B.ThicknessConstraint(0.25, "max")),
                     ("point", B.PointConstraint(0.3, 4)),
                     ("point_pv", B.PointConstraint(-0.1, 2, parent_vertex=6))):
        hs = [Holder(g["con_a"]), Holder(g["con_b"])]
        con.make(1, hs)()
        _close(hs[1].parameters, g["con_" + tag], msg=tag)
        _close(hs[0].parameters, g["con_a"], msg=tag + " (parent untouched)")
    h0 = [Holder(g["con_a"]), Holder(g["con_b"])]
    B.ThicknessConstraint(0.1, "min").make(0, h0)()
    _close(h0[0].parameters, g["con_first"], msg="first surface against zero")
    hz = Holder(g["con_b"])
    B.ThicknessConstraint(0.05, "max", parent="zero").make(hz, None)()
    _close(hz.parameters, g["con_zero"], msg="parent=zero")
    hc = Holder(g["con_a"])
    B.ClipConstraint(-0.5, 0.7).make(hc, None)()
    _close(hc.parameters, g["con_clip"], msg="clip")

    # ---- vector generators
    zero = torch.tensor(P, device=dev)
    _close(B.SecondSurfaceVG(g["vg_second_points"]).generate(zero), g["vg_second"], msg="SecondSurfaceVG")
    _close(B.FromPointVG((-4.0, 0.1, -0.05)).generate(zero), g["vg_point"], msg="FromPointVG")
    _close(B.FromVectorVG((0.3, -0.4, 1.2)).generate(zero), g["vg_vector"], msg="FromVectorVG")
    _close(B.FromAxisVG((-3.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0)).generate(zero), g["vg_axis"],
           msg="FromAxisVG")

    # ---- 2-D parametric segments (these DO construct here: the reference's cannot, at HEAD)
    zd = D.ManualBasePointDistribution(2, points=g["seg_zero"])
    od = D.ManualBasePointDistribution(2, points=g["seg_one"])
    for flip in (False, True):
        sb = B.ParametricSegmentBoundary(zd, od, flip_norm=flip, initial_parameters=g["seg_init"])
        sb.update()
        fs = torch.stack([sb[f] for f in SEG], 1)
        _close(fs, g[f"seg_fields_{int(flip)}"], msg=f"segments flip={flip}")
        (grad,) = torch.autograd.grad((fs * torch.tensor(g["seg_w"], device=dev)).sum(), [sb.parameters])
        _close(grad, g[f"seg_grad_{int(flip)}"], 1e-13, f"segment gradient flip={flip}")
    msb = B.ParametricMultiSegmentBoundary(
        zd, od, [B.ThicknessConstraint(0.0, "min"), B.ThicknessConstraint(0.15, "min")], [True, False],
        initial_parameters=[g["mseg_init0"], g["mseg_init1"]])
    msb.update()
    _close(msb.surfaces[0].parameters, g["mseg_p0"], msg="multi-segment p0")
    _close(msb.surfaces[1].parameters, g["mseg_p1"], msg="multi-segment p1")
    _close(torch.stack([msb[f] for f in SEG], 1), g["mseg_fields"], msg="multi-segment fields")

    # ---- distributions and the 3-D AperatureSource (values AND order)
    _close(D.StaticUniformCircle(11, 0.2).points, g["dist_circle"], msg="StaticUniformCircle")
    _close(D.StaticUniformBeam(-1.5, 1.5, 10).points, g["dist_beam"], msg="StaticUniformBeam")
    _close(D.StaticUniformAngularDistribution(-0.1, 0.25, 7).angles, g["dist_angles"], msg="angles")
    a = D.StaticUniformCircle(11, 0.2)
    D.BasePointTransformation(a, translation=(-10, 0, 0))
    b2 = D.StaticUniformCircle(11, 0.9)
    D.BasePointTransformation(b2)
    src = S.AperatureSource(3, a, b2, [575.0], dense=False,
                            extra_fields={"object_coords": ("start_point", a, "points")})
    for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength", "object_coords"):
        _close(src[f], g["ap_undense_" + f], msg="undense " + f)
    a = D.StaticUniformCircle(4, 0.2)
    D.BasePointTransformation(a, translation=(-3, 0, 0))
    b2 = D.StaticUniformCircle(5, 0.9)
    D.BasePointTransformation(b2)
    src = S.AperatureSource(3, a, b2, [450.0, 650.0], dense=True,
                            extra_fields={"end_tag": ("end_point", np.arange(5, dtype=np.float64) * 10)})
    for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength", "end_tag"):
        _close(src[f], g["ap_dense_" + f], msg="dense " + f)


def test_host_classes_reproduce_the_reference_source_on_the_cpu(cpu_backend):
    _check_all()


@pytest.mark.gpu
def test_host_classes_reproduce_the_reference_source_on_the_device():
    import tensorflowraytrace_amd as tfa
    tfa.set_device("cuda:0")
    _check_all()
