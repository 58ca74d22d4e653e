"""
Mesh / parametrisation tooling and the cylindrical light guide against
tests/golden/reference_mesh.npz -- outputs of the reference's OWN tfrt/mesh_tools.py and
tfrt/boundaries.py executed under tests/tf_shim (tests/golden/make_reference_mesh_golden.py).
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_mesh.npz")
TRI = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2")


def _same_mesh(mesh, points, faces, what):
    """Same triangles over the same points.  Vertex numbering and face order are free (they are
    bookkeeping of the generator), so compare the set of faces as coordinate triples."""
    assert mesh.points.shape == points.shape, what
    got = np.sort(np.round(mesh.points[mesh.triangles()].reshape(-1, 9), 12), axis=0)
    want = np.sort(np.round(points[faces.reshape(-1, 4)[:, 1:]].reshape(-1, 9), 12), axis=0)
    # (rows as wholes: lexicographic order of the 9-tuples)
    key = lambda a: a[np.lexsort(a.T[::-1])]
    g = key(np.round(mesh.points[mesh.triangles()].reshape(-1, 9), 12))
    w = key(np.round(points[faces.reshape(-1, 4)[:, 1:]].reshape(-1, 9), 12))
    np.testing.assert_allclose(g, w, atol=1e-12, err_msg=what)


def test_mesh_generators_and_parametrisation_tools_equal_the_reference():
    import tensorflowraytrace_amd.mesh_tools as mt
    g = np.load(GOLD)
    h = mt.hexagonal_mesh(radius=1.0, step_count=4)
    np.testing.assert_allclose(h.points, g["hex_points"], atol=1e-14)       # same order, too
    assert np.array_equal(h.faces, g["hex_faces"])
    c = mt.circular_mesh(1.0, 0.3)
    np.testing.assert_allclose(c.points, g["circ_points"], atol=1e-14)
    assert np.array_equal(c.faces, g["circ_faces"])
    cyl = mt.cylindrical_mesh((0.0, 0.0, 0.0), (0.0, 0.0, 3.0), radius=0.5, theta_res=8, z_res=5,
                              start_cap=True, end_cap=True)
    np.testing.assert_allclose(cyl.points, g["cyl_points"], atol=1e-14)
    assert np.array_equal(cyl.faces, g["cyl_faces"])

    top = mt.get_closest_point(h, (0.0, 0.0, 0.0))
    assert top == int(g["hex_top"])
    vmap, acc = mt.mesh_parametrization_tools(h, top)
    assert np.array_equal(np.asarray(vmap), g["hex_vmap"])
    np.testing.assert_allclose(np.asarray(acc, dtype=np.float64), g["hex_acc"], atol=0)
    np.testing.assert_allclose(np.asarray(mt.mesh_smoothing_tool(h, [4, 2, 1]), dtype=np.float64),
                               g["hex_smoother"], atol=1e-15)
    assert [len(x) for x in mt.find_generations(top, h)] == list(g["hex_generation_sizes"])
    np.testing.assert_allclose(np.asarray(mt.gaussian_weights(1.5, 5)), g["gauss"], atol=1e-15)
    bump = mt.PolyData(g["flat_in"], h.faces)
    np.testing.assert_allclose(np.asarray(mt.get_flat_initial(bump, axis=2)), g["flat_initial"], atol=0)
    np.testing.assert_allclose(bump.points, g["flat_points_after"], atol=0)


def _guide(sym):
    import tensorflowraytrace_amd.boundaries as B
    g = B.ParametricCylindricalGuide(
        (0.0, 0.0, 0.0), (0.0, 0.0, 5.0), 0.5, theta_res=12, z_res=6, start_cap=False, end_cap=False,
        rotationally_symmetric=sym, initial_taper=(0.05, 0.25))
    if not sym:
        with torch.no_grad():
            k = torch.arange(g.parameters.shape[0], dtype=torch.float64, device=g.parameters.device)
            g.parameters.add_(0.01 * torch.sin(1.7 * k))
    g.update()
    return g


def _check_guides():
    gold = np.load(GOLD)
    for sym in (True, False):
        tag = "sym" if sym else "full"
        g = _guide(sym)
        dev = g.parameters.device
        np.testing.assert_allclose(g.parameters.detach().cpu().numpy(), gold[f"guide_{tag}_params"], atol=1e-15)
        fv = torch.stack([g[k] for k in TRI], 1)
        np.testing.assert_allclose(fv.detach().cpu().numpy(), gold[f"guide_{tag}_fields"], atol=1e-14)
        np.testing.assert_allclose(g["norm"].detach().cpu().numpy(), gold[f"guide_{tag}_norm"], atol=1e-13)
        loss = (fv * torch.tensor(gold["guide_w"], device=dev)).sum() + \
            (g["norm"] * torch.tensor(gold["guide_wn"], device=dev)).sum()
        (grad,) = torch.autograd.grad(loss, [g.parameters])
        want = gold[f"guide_{tag}_grad"]
        assert np.abs(grad.cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max(), tag
        np.testing.assert_allclose(np.asarray(g.accumulator, dtype=np.float64),
                                   gold[f"guide_{tag}_accumulator"], atol=0)


def test_cylindrical_guide_equals_the_reference_on_the_cpu(cpu_backend):
    _check_guides()


@pytest.mark.gpu
def test_cylindrical_guide_equals_the_reference_on_the_device():
    import tensorflowraytrace_amd as tfa
    tfa.set_device("cuda:0")
    _check_guides()
