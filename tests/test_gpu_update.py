"""Parameter-side kernels of the optimiser step (tfrt_sgd_process, tfrt_csr_matvec) against the
same arithmetic as eager float64/float32 torch ops (optimizer.py:223-257, 277-282, 316)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_process(g, scale, clip):
    g = torch.where(torch.isfinite(g), g, torch.zeros_like(g))
    return torch.clamp(g * scale, -clip, clip)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [0, 1, 271, 5167, 100_003])
def test_sgd_process_matches_eager_ops(dtype, n):
    from tensorflowraytrace_amd import ops
    gen = torch.Generator().manual_seed(n + 7)
    g = torch.randn(n, generator=gen, dtype=torch.float64) * 3e-3
    if n > 10:
        g[1], g[5], g[7] = float("nan"), float("inf"), float("-inf")
        g[9] = 1e30
    g = g.to(dtype).cuda()
    p = torch.randn(n, generator=gen, dtype=torch.float64).to(dtype).cuda()
    scale, clip, lr = 0.37, 1e-3, 0.01
    want = _ref_process(g, scale, clip)
    got = ops.sgd_process(g, scale, clip)
    assert torch.equal(got, want)                      # same unfused arithmetic: bit-identical
    p2 = p.clone()
    got2 = ops.sgd_process(g, scale, clip, param=p2, sgd_learning_rate=lr)
    assert torch.equal(got2, want)
    want_p = p - torch.as_tensor(lr, dtype=dtype).cuda() * want
    assert torch.equal(p2, want_p)


def test_sgd_process_multi_equals_the_per_tensor_launches():
    """tfrt_sgd_process_multi through ctypes: several parameter tensors in one launch, scalars from
    a device table with one row per tensor; bit-identical to tfrt_sgd_process on each."""
    import ctypes
    from tensorflowraytrace_amd import _lib, ops
    L = _lib.lib()
    sizes = [5167, 271, 1, 0, 100_003]
    gen = torch.Generator().manual_seed(5)
    grads, params, want_p, want_g = [], [], [], []
    rows = []
    for k, n in enumerate(sizes):
        g = torch.randn(n, generator=gen, dtype=torch.float64) * 3e-3
        if n > 10:
            g[1], g[5], g[7], g[9] = float("nan"), float("inf"), float("-inf"), 1e30
        p = torch.randn(n, generator=gen, dtype=torch.float64)
        scale, clip, lr = 0.37 + 0.1 * k, 1e-3 * (k + 1), 0.01 / (k + 1)
        rows.append([scale, clip, lr])
        grads.append(g.cuda())
        params.append(p.cuda())
        pg = _ref_process(grads[-1], scale, clip)
        want_g.append(pg)
        want_p.append(params[-1] - torch.as_tensor(lr, dtype=torch.float64).cuda() * pg)
    hyper = torch.tensor(rows, dtype=torch.float64).cuda()
    k = len(sizes)
    processed = [torch.full_like(g, 7.0) for g in grads]
    arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() if t.numel() else None for t in ts])
    nn = (ctypes.c_int64 * k)(*sizes)
    _lib.check(L.tfrt_sgd_process_multi(k, arr(grads), arr(processed), arr(params), nn,
                                        ops._p(hyper), ops._stream(hyper)), "tfrt_sgd_process_multi")
    for i in range(k):
        assert torch.equal(processed[i], want_g[i]), i
        assert torch.equal(params[i], want_p[i]), i
    # more tensors than one launch takes, and a missing table
    assert L.tfrt_sgd_process_multi(9, arr(grads), None, None, nn, ops._p(hyper), None) == -1
    assert L.tfrt_sgd_process_multi(k, arr(grads), None, None, nn, None, None) == -1


def test_sgd_process_rejects_bad_arguments():
    from tensorflowraytrace_amd import ops
    from tensorflowraytrace_amd._lib import TfrtError
    g = torch.zeros(8, dtype=torch.float64).cuda()
    with pytest.raises(TfrtError):
        ops.sgd_process(g.to(torch.float16), 1.0, 1.0)
    with pytest.raises(TfrtError):
        ops.sgd_process(g, 1.0, 1.0, param=torch.zeros(9, dtype=torch.float64).cuda())
    with pytest.raises(TfrtError):
        ops.sgd_process(g, 1.0, -1.0)
    with pytest.raises(TfrtError):
        ops.sgd_process(torch.zeros(8, dtype=torch.float64), 1.0, 1.0)   # CPU tensor


@pytest.mark.parametrize("n,per_row", [(1, 1), (271, 7), (5167, 7), (600, 200)])
def test_csr_matvec_matches_dense_product(n, per_row):
    from tensorflowraytrace_amd import ops
    rng = np.random.default_rng(n)
    a = np.zeros((n, n))
    for r in range(n):
        cols = rng.choice(n, size=min(per_row, n), replace=False)
        a[r, cols] = rng.standard_normal(cols.size)
    if n > 3:
        a[2, :] = 0.0                                   # an empty row
    x = torch.tensor(rng.standard_normal(n)).cuda()
    m = ops.CsrMatrix(a, x.device)
    assert m.nnz == int((a != 0).sum())
    y = m.matvec(x)
    want = a @ x.cpu().numpy()
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-13, atol=1e-13)
    y2 = m.matvec(x.reshape(-1, 1))
    assert y2.shape == (n, 1)
    np.testing.assert_allclose(y2.cpu().numpy()[:, 0], want, rtol=1e-13, atol=1e-13)


def test_optimizer_step_with_accumulator_and_smoother_matches_dense_host_math():
    """single_step + smooth on the GPU (CSR kernels, fused update) vs the same step written
    out with dense float64 matrices from the gradients the optimiser itself reports."""
    from test_gpu_engine import _build_lens
    import tfrt.optimizer as optimizer
    eng, system, lens, target, source = _build_lens(4000, k=5)
    params = lens.parameters

    def erf(engine):
        fin = engine.finished_rays
        return (fin["y_end"] ** 2 + fin["z_end"] ** 2)

    rng = np.random.default_rng(3)
    P = params[0].numel()
    acc = np.eye(P) + (rng.random((P, P)) < 0.05) * 0.3
    smo = np.eye(P) * 0.8 + np.roll(np.eye(P), 1, axis=1) * 0.2
    opt = optimizer.SGD_Optimizer(eng, params, erf, trace_depth=3, learning_rate=1e-3,
                                  grad_clip=1e-2, speculative=False)
    opt.suppress_warnings = True
    raw, _, _ = opt.raw_gradient()      # runs the constraints: read the parameters afterwards
    p_before = [p.detach().clone() for p in params]
    want = []
    for i, (g, p0) in enumerate(zip(raw, p_before)):
        g = torch.where(torch.isfinite(g), g, torch.zeros_like(g)) * 1e-3
        g = torch.clamp(g, -1e-2, 1e-2)
        if i == 0:
            g = (torch.tensor(acc).cuda() @ g.reshape(-1, 1)).reshape(g.shape)
        want.append(p0 - 0.01 * g)
    want[0] = (torch.tensor(smo).cuda() @ want[0].reshape(-1, 1)).reshape(want[0].shape)
    err = opt.single_step([acc] + [None] * (len(params) - 1))
    optimizer.SGD_Optimizer.smooth(params[0], smo)
    assert float(err) > 0
    for p, w in zip(params, want):
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.cpu().numpy(), rtol=1e-12, atol=1e-14)


# ------------------------------------------------------------------------------------------
# parameters -> faces in one launch (tfrt_param_faces_*, boundaries.py:1065-1092 + 890-923)

def _param_surface(k, flip, seed):
    import tensorflowraytrace_amd.mesh_tools as mt
    mesh = mt.hexagonal_mesh(1.0, k)
    mesh.rotate_y(90)
    rng = np.random.default_rng(seed)
    V = mesh.n_points
    zero = torch.tensor(mesh.points).cuda()
    vec = torch.tensor(rng.standard_normal((V, 3))).cuda()
    faces = torch.tensor(mesh.triangles()[:, ::-1].copy() if flip else mesh.triangles(),
                         dtype=torch.int32).cuda()
    p = torch.tensor(rng.uniform(-0.2, 0.2, V)).cuda()
    return zero, vec, faces, p


@pytest.mark.parametrize("k,flip,masked", [(1, False, False), (7, True, True), (24, False, True)])
def test_param_faces_is_bit_identical_with_the_two_step_path_and_matches_its_gradient(k, flip, masked):
    from tensorflowraytrace_amd import ops
    zero, vec, faces, p0 = _param_surface(k, flip, 11 * k)
    F = faces.shape[0]
    mask = None
    if masked:
        mask = (torch.rand(F, 3, generator=torch.Generator().manual_seed(k)) < 0.6).to(torch.uint8).cuda()
    p1 = p0.clone().requires_grad_(True)
    fv1, n1 = ops.param_faces(p1, zero, vec, faces, mask)
    p2 = p0.clone().requires_grad_(True)
    fv2, n2 = ops.build_faces(zero + p2.reshape(-1, 1) * vec, faces, mask)
    assert torch.equal(fv1, fv2) and torch.equal(n1, n2)
    gen = torch.Generator().manual_seed(5)
    w_fv = torch.randn(F, 9, generator=gen, dtype=torch.float64).cuda()
    w_n = torch.randn(F, 3, generator=gen, dtype=torch.float64).cuda()
    for use_fv, use_n in ((True, False), (False, True), (True, True)):
        def loss(fv, n):
            out = 0.0
            if use_fv:
                out = out + (fv * w_fv).sum()
            if use_n:
                out = out + (n * w_n).sum()
            return out
        g1, = torch.autograd.grad(loss(fv1, n1), p1, retain_graph=True)
        g2, = torch.autograd.grad(loss(fv2, n2), p2, retain_graph=True)
        scale = float(g2.abs().max())
        assert float((g1 - g2).abs().max()) <= 1e-13 * max(scale, 1.0)
    if masked:      # corners the map switches off take no gradient at all
        only = torch.zeros(F, 9, dtype=torch.float64).cuda()
        only[:, :3] = (1 - mask[:, :1].double())
        g, = torch.autograd.grad((fv1 * only).sum(), p1, retain_graph=True)
        assert float(g.abs().max()) == 0.0


def test_param_faces_rejects_mismatched_shapes_and_handles_empty_meshes():
    from tensorflowraytrace_amd import ops
    from tensorflowraytrace_amd._lib import TfrtError
    zero, vec, faces, p = _param_surface(2, False, 3)
    with pytest.raises(TfrtError):
        ops.param_faces(p[:-1], zero, vec, faces)
    with pytest.raises(TfrtError):
        ops.param_faces(p, zero, vec[:-1], faces)
    fv, n = ops.param_faces(p, zero, vec, faces[:0])
    assert fv.shape == (0, 9) and n.shape == (0, 3)


def test_parametric_boundary_uses_the_fused_update_and_serves_vertices_on_demand():
    import tensorflowraytrace_amd.boundaries as boundaries
    import tensorflowraytrace_amd.mesh_tools as mt
    from tensorflowraytrace_amd import ops
    zp = mt.hexagonal_mesh(1.0, 6)
    zp.rotate_y(90)
    surf = boundaries.ParametricTriangleBoundary(
        zp, boundaries.FromVectorVG((1, 0, 0)), initial_parameters=0.0,
        material_dict={"mat_in": 1, "mat_out": 0})
    with torch.no_grad():
        surf.parameters.copy_(torch.linspace(-0.1, 0.2, surf.parameters.shape[0]).cuda())
    surf.update()
    assert surf.__dict__.get("_vertices_pending") is True        # nothing formed (V,3) yet
    verts = surf.vertices
    want = surf._zero_points + surf.parameters.reshape(-1, 1) * surf.vectors
    assert torch.equal(verts, want) and verts.requires_grad
    faces = torch.as_tensor(surf.faces[:, 1:].astype(np.int32)).cuda()
    fv, nrm = ops.build_faces(want, faces)
    assert torch.equal(surf.face_verts, fv) and torch.equal(surf["norm"], nrm)
    assert torch.equal(surf["x1"], fv[:, 3])
    g, = torch.autograd.grad(surf.face_verts.sum() + surf["norm"][:, 0].sum(), surf.parameters)
    g2, = torch.autograd.grad(fv.sum() + nrm[:, 0].sum(), surf.parameters)
    assert float((g - g2).abs().max()) < 1e-13
    surf.update_mesh_from_vertices()
    np.testing.assert_array_equal(surf.mesh.points, want.detach().cpu().numpy())


def test_param_faces_of_several_surfaces_in_one_launch_equal_the_single_launches():
    """ops.ParamFacesBatch / tfrt_param_faces_*_multi: two parametric surfaces and a fixed one
    between them -> one merged block; rows, normals and parameter gradients (through the block,
    through a boundary's own rows and through the normals) equal those of tfrt_param_faces_*."""
    from tensorflowraytrace_amd import ops

    class Holder:      # stands for a boundary: the batch writes _face_verts / _norm into it
        pass

    za, va, fa, pa0 = _param_surface(7, True, 3)
    zb, vb, fb, pb0 = _param_surface(12, False, 4)
    mask_b = (torch.rand(fb.shape[0], 3, generator=torch.Generator().manual_seed(2)) < 0.7).to(torch.uint8).cuda()
    fixed = Holder()
    fixed._face_verts = torch.randn(5, 9, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).cuda()
    fixed.__dict__["_face_verts_value"] = fixed._face_verts
    pa, pb = pa0.clone().requires_grad_(True), pb0.clone().requires_grad_(True)
    a, b = Holder(), Holder()
    batch = ops.ParamFacesBatch()
    batch.add(a, pa, za, va, fa, None)
    batch.add(b, pb, zb, vb, fb, mask_b)
    batch.flush([a, fixed, b])
    block, rows = batch.merged
    Fa, Fb = fa.shape[0], fb.shape[0]
    assert [(r0, r1) for _, r0, r1 in rows] == [(0, Fa), (Fa, Fa + 5), (Fa + 5, Fa + 5 + Fb)]
    qa, qb = pa0.clone().requires_grad_(True), pb0.clone().requires_grad_(True)
    fva, na = ops.param_faces(qa, za, va, fa, None)
    fvb, nb = ops.param_faces(qb, zb, vb, fb, mask_b)
    assert torch.equal(block, torch.cat([fva, fixed._face_verts, fvb]))
    assert torch.equal(a._face_verts, fva) and torch.equal(b._face_verts, fvb)
    assert torch.equal(a._norm, na) and torch.equal(b._norm, nb)
    gen = torch.Generator().manual_seed(9)
    w = torch.randn(block.shape, generator=gen, dtype=torch.float64).cuda()
    wn = torch.randn(Fb, 3, generator=gen, dtype=torch.float64).cuda()
    loss1 = (block * w).sum() + (b._norm * wn).sum() + (a._face_verts ** 2).sum()
    loss2 = ((fva * w[:Fa]).sum() + (fvb * w[Fa + 5:]).sum() + (nb * wn).sum() + (fva ** 2).sum())
    g1 = torch.autograd.grad(loss1, [pa, pb])
    g2 = torch.autograd.grad(loss2, [qa, qb])
    for x, y in zip(g1, g2):
        assert float((x - y).abs().max()) <= 1e-13 * max(float(y.abs().max()), 1.0)


def test_optical_system_update_runs_the_face_updates_as_one_launch_and_needs_no_concatenation():
    """OpticalSystem3D.update(): both lens surfaces hand their update to the batch, the target's
    faces are copied into place, and the merged face tensor IS that block; boundaries keep their
    own rows; the same scene updated boundary by boundary (no batch) gives the same faces."""
    import bench
    eng, system, params = bench.build_scene(4096, 9, 5, torch.float64)
    system.update()
    merged = system._merged_face_verts
    parts = [b.face_verts for lst in (system._optical, system._stop, system._target) for b in lst if bool(b)]
    assert torch.equal(merged, torch.cat(parts))
    lens = [b for b in system._optical if hasattr(b, "parameters")]
    assert len(lens) == 2
    for b in lens:   # rows of the block, not copies
        assert b.face_verts.untyped_storage().data_ptr() == merged.untyped_storage().data_ptr()
    # boundary by boundary
    for b in lens:
        b.update()
    for b, rows in zip(lens, (parts[0], parts[1])):
        assert torch.equal(b.face_verts, rows)
    g = torch.autograd.grad((merged ** 2).sum(), params, allow_unused=True)
    assert all(x is not None and bool(torch.isfinite(x).all()) and float(x.abs().max()) > 0 for x in g)
