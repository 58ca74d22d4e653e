"""tfrt_ray_order / tfrt_permute_rays / tfrt_gather_rows / tfrt_restore_order (csrc/tfrt_order.hip):
the device side of the ray-order contract of OpticalEngine.ray_trace (tfrt/engine.py:2311-2330,
2069-2111, 1379-1403)."""
import numpy as np
import pytest
import torch

import scene_util
import order_reference

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _u32(t):
    return t.long() & 0xFFFFFFFF


def _lens_rays(n, dtype=torch.float32, seed=None):
    scene = scene_util.lens_scene(n, k_front=6, k_back=4, seed=seed)
    return torch.tensor(scene["rays"], dtype=dtype, device=DEV)


@pytest.mark.parametrize("n", [1, 63, 64, 4097, 70001, 300 * 1024 + 5, 1000000, 4 * 2 ** 20 + 77])
def test_order_is_the_stable_argsort_of_its_keys(n):
    """Every tile size of the radix sort (4, 8 and 16 items per thread), partial tiles, one ray."""
    from tensorflowraytrace_amd import ops
    rays = _lens_rays(n, seed=3)
    perm, keys = ops.ray_order(rays, return_keys=True)
    k = _u32(keys)
    want = torch.argsort(k, stable=True).int()
    assert torch.equal(perm, want)
    again = ops.ray_order(rays)
    assert torch.equal(again, perm)                 # the same on every run


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
def test_keys_are_hilbert_indices_of_the_lines_foot_points(dtype):
    from tensorflowraytrace_amd import ops
    n = 50000
    rays = _lens_rays(n, dtype, seed=11)
    perm, keys = ops.ray_order(rays, axis=(1.0, 0.2, -0.1), return_keys=True)
    want, bits = order_reference.keys(rays, (1.0, 0.2, -0.1))
    got = _u32(keys).cpu()
    # float32 coordinates at a cell border may fall either way: a handful of rays, by one cell
    assert float((got != want).float().mean()) < 2e-3
    assert int(got.max()) < (1 << (2 * bits))
    # neighbours in the order are neighbours in space: the 64-ray groups are small patches
    e = rays[3:].double()[:, perm.long()].cpu()
    groups = e[1:, : n // 64 * 64].reshape(2, -1, 64)
    extent = (groups.max(2).values - groups.min(2).values).max(0).values
    assert float(extent.median()) < 0.15            # aperture radius 0.98: a random group spans ~1.9


def test_order_with_scene_faces_sampled_frame_and_isotropic_rays():
    from tensorflowraytrace_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 30000
    d = torch.randn(3, n, generator=g, dtype=torch.float64)
    d /= d.norm(dim=0)
    s = torch.zeros(3, n, dtype=torch.float64)
    rays = torch.cat([s, d]).to(DEV)
    fv = torch.randn(500, 9, generator=g, dtype=torch.float64).to(DEV)
    perm, keys = ops.ray_order(rays, fv, return_keys=True)
    assert torch.equal(perm, torch.argsort(_u32(keys), stable=True).int())
    # isotropic: ordered by direction -- neighbours in the order point the same way
    dd = d[:, perm.long().cpu()]
    cos = (dd[:, :-1] * dd[:, 1:]).sum(0)
    assert float(cos.median()) > 0.99


def test_rays_that_are_no_lines_come_last_and_equal_rays_keep_their_order():
    from tensorflowraytrace_amd import ops
    n = 10000
    rays = _lens_rays(n, torch.float64)
    rays[:, 17] = float("nan")
    rays[3:, 4000] = rays[:3, 4000]                 # zero length
    rays[:, 9000] = float("inf")
    perm = ops.ray_order(rays)
    assert sorted(perm[-3:].tolist()) == [17, 4000, 9000]
    same = rays[:, :1].repeat(1, 5000).contiguous()
    assert torch.equal(ops.ray_order(same), torch.arange(5000, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
def test_permute_rays_and_gather_rows(dtype):
    from tensorflowraytrace_amd import ops
    n = 123457
    g = torch.Generator().manual_seed(2)
    rays = torch.randn(6, n, generator=g).to(dtype).to(DEV)
    perm = torch.randperm(n, generator=g).int().to(DEV)
    assert torch.equal(ops.permute_rays(rays, perm), rays[:, perm.long()])
    wide = torch.randn(6, n + 100, generator=g).to(dtype).to(DEV)[:, :n]      # row stride > n
    assert torch.equal(ops.permute_rays(wide, perm), wide[:, perm.long()])
    for t in (torch.randn(3, n, generator=g, dtype=torch.float64),
              torch.randint(0, 1 << 30, (n,), generator=g, dtype=torch.int32),
              torch.randint(0, 255, (2, n), generator=g, dtype=torch.uint8),
              torch.randn(2, n, generator=g).half()):
        t = t.to(DEV)
        assert torch.equal(ops.gather_rows(t, perm), t[..., perm.long()])
    # a device-side count bounds the rows that are written
    out = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    idx = torch.arange(n, dtype=torch.int32, device=DEV)
    ops.gather_rows(idx, perm, torch.tensor([1000], dtype=torch.int32, device=DEV), out=out)
    assert torch.equal(out[:1000], perm[:1000]) and bool((out[1000:] == -1).all())


def _trace_pair(n_rays, dtype, passes=4):
    from tensorflowraytrace_amd import ops, _lib
    from test_gpu_trace3d import _gpu_scene
    scene = scene_util.lens_scene(n_rays, k_front=10, k_back=5, seed=9)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src, fv, sc, params = _gpu_scene(scene, dtype, cluster="group")
    return scene, src, fv, sc, params, flags


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_trace_over_an_ordered_source_returns_the_reference_order(dtype):
    """trace3d(perm=...): the classes come back restored inside the autograd node -- every ray set
    bit for bit what the natural-order trace gives, and the parameter gradients through the
    restored rows equal the natural-order ones."""
    from tensorflowraytrace_amd import ops
    scene, src, fv, sc, (p_f, p_b), flags = _trace_pair(60000, dtype)
    ref = ops.trace3d(src, fv, sc, max_passes=4, flags=flags)

    def loss(o):
        fin = o["finished"]
        goal = torch.tensor(scene["goal"], dtype=torch.float64, device=DEV)[o["finished_id"].long()]
        act = o["active"]
        return (((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
                + 1e-3 * (act[3].double() * torch.arange(act.shape[1], device=DEV) / act.shape[1]).sum())
    g_ref = torch.autograd.grad(loss(ref), [p_f, p_b], retain_graph=True)
    perm = ops.ray_order(src, fv)
    sc2 = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                          n_table=ops.gather_rows(sc.n_table, perm),
                          face_grad_mask=sc.face_grad_mask, cluster_order=sc.cluster_order,
                          coherent_rays=True)
    out = ops.trace3d(ops.permute_rays(src, perm), fv, sc2, max_passes=4, flags=flags, perm=perm)
    assert np.array_equal(out["counts"], ref["counts"])
    for cls in ("finished", "active", "dead", "stopped", "unfinished"):
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), cls
        if cls != "unfinished":
            assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), cls
        assert torch.equal(out[cls], ref[cls]), cls
    g = torch.autograd.grad(loss(out), [p_f, p_b])
    tol = 1e-11 if dtype == torch.float64 else 2e-6
    for a, b in zip(g, g_ref):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())


def test_restore_order_of_rows_that_span_many_passes():
    """A long trace in a closed box of mirrors: 40 passes, every pass non-empty, the active history
    holds every ray 40 times -- the (pass, original id) bitmap has 40 segments."""
    from tensorflowraytrace_amd import ops, _lib
    from test_gpu_trace3d import _soup_scene
    rays, fv, scene = _soup_scene(5, 800, 6000)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    ref = ops.trace3d(rays, fv, scene("group"), max_passes=40, flags=flags)
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(rays.shape[1], generator=g).int().to(DEV)
    sc = scene("group")
    sc.coherent_rays = True
    if sc.n_table is not None:
        sc.n_table = ops.gather_rows(sc.n_table, perm)
    out = ops.trace3d(ops.permute_rays(rays, perm), fv, sc, max_passes=40, flags=flags, perm=perm)
    assert np.array_equal(out["counts"], ref["counts"])
    for cls in ("finished", "active", "dead", "stopped", "unfinished"):
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), cls
        assert torch.equal(out[cls], ref[cls]), cls


def _run_radius(cent, run):
    n = cent.shape[0] // run * run
    c = cent[:n].reshape(-1, run, 3)
    return np.linalg.norm(c - c.mean(1, keepdims=True), axis=2).max(1)


@pytest.mark.parametrize("k,extra", [(3, 0), (20, 2), (41, 2), (120, 0)])
def test_device_cluster_order_against_the_numpy_kd_order(k, extra):
    """tfrt_cluster_order: a permutation; aligned runs of 16 / 128 faces as compact as the host
    k-d order's (mean bounding radius within 10 %); outsized faces (a target plane) last."""
    from tensorflowraytrace_amd import ops
    import cluster_reference
    import tfrt.mesh_tools as mt
    mesh = mt.hexagonal_mesh(1.0, k)
    tri = np.asarray(mesh.points[mesh.triangles()].reshape(-1, 9), dtype=np.float64)
    tri[:, 2::3] += 0.15 * (tri[:, 0::3] ** 2 + tri[:, 1::3] ** 2)          # a curved surface
    if extra:
        big = np.array([[-50, -50, 10, 50, -50, 10, 50, 50, 10], [-50, -50, 10, 50, 50, 10, -50, 50, 10.0]])
        tri = np.concatenate([tri[:100], big, tri[100:]])                   # (in the middle of the list)
    fv = torch.tensor(tri, device=DEV)
    order = ops.cluster_order(fv)
    assert order.dtype == torch.int32 and order.is_cuda
    o = order.cpu().numpy()
    M = tri.shape[0]
    assert np.array_equal(np.sort(o), np.arange(M))
    assert torch.equal(ops.cluster_order(fv), order)                        # the same on every run
    if extra:
        assert sorted(o[-2:].tolist()) == [100, 101]
    ref = cluster_reference.cluster_order_numpy(torch.tensor(tri)).numpy()
    cent = tri.reshape(-1, 3, 3).mean(1)
    n_small = M - extra
    for run in (16, 128):
        if n_small < 4 * run:
            continue
        got, want = _run_radius(cent[o[:n_small]], run).mean(), _run_radius(cent[ref[:n_small]], run).mean()
        assert got <= 1.10 * want, (run, got, want)


def test_device_cluster_order_tiny_and_degenerate_inputs():
    from tensorflowraytrace_amd import ops
    g = torch.Generator().manual_seed(0)
    for M in (0, 1, 5, 16, 17, 129):
        fv = torch.randn(M, 9, generator=g, dtype=torch.float64).to(DEV)
        o = ops.cluster_order(fv).cpu().numpy()
        assert np.array_equal(np.sort(o), np.arange(M))
    same = torch.ones(1000, 9, dtype=torch.float64, device=DEV)            # every centroid equal
    assert np.array_equal(np.sort(ops.cluster_order(same).cpu().numpy()), np.arange(1000))
    bad = torch.randn(300, 9, generator=g, dtype=torch.float64)
    bad[7] = float("nan")
    bad[9, 0] = float("inf")
    o = ops.cluster_order(bad.to(DEV)).cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(300))
