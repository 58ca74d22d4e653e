"""
GPU parity of the 3-D hot path (through the C ABI) against the float64 oracle.

Tolerances (BASELINE.json north_star): 1e-5 relative on ray endpoints and gradients for
float32 ray state; the float64-state path is held to 1e-9.
"""
import numpy as np
import pytest
import torch

import scene_util
import oracle_util
from oracle import tracer

pytestmark = pytest.mark.gpu

NAMES = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


def _gpu_scene(scene, dtype, p_f=None, p_b=None, masks=(None, None), cluster=False):
    from tensorflowraytrace_amd import ops
    dev = torch.device("cuda:0")
    tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt, device=dev)
    p_f = tt(scene["p_f"] if p_f is None else p_f).requires_grad_(True)
    p_b = tt(scene["p_b"] if p_b is None else p_b).requires_grad_(True)
    vec = tt(scene["vector"]).reshape(1, 3)
    v_f = tt(scene["zero_f"]) + p_f.reshape(-1, 1) * vec
    v_b = tt(scene["zero_b"]) + p_b.reshape(-1, 1) * vec
    m = [None if x is None else tt(x, torch.uint8) for x in masks]
    fv_f, _ = ops.build_faces(v_f, tt(scene["faces_f"], torch.int32), m[0])
    fv_b, _ = ops.build_faces(v_b, tt(scene["faces_b"], torch.int32), m[1])
    fv_t, _ = ops.build_faces(tt(scene["target_verts"]), tt(scene["target_faces"], torch.int32))
    fv = torch.cat([fv_f, fv_b, fv_t])
    nf, nb, nt = fv_f.shape[0], fv_b.shape[0], fv_t.shape[0]
    cat = torch.cat([torch.zeros(nf + nb, dtype=torch.int32), torch.full((nt,), 2, dtype=torch.int32)]).to(dev)
    mat_in = torch.cat([torch.ones(nf + nb, dtype=torch.int32), torch.zeros(nt, dtype=torch.int32)]).to(dev)
    mat_out = torch.zeros(nf + nb + nt, dtype=torch.int32, device=dev)
    wl = torch.tensor(scene["wavelength"], dtype=torch.float64)
    n_table = torch.stack([tracer.MATERIALS["vacuum"](wl), tracer.MATERIALS["acrylic"](wl)]).to(dev)
    gmask = (cat == 0).to(torch.uint8)  # the target plane is a constant
    # cluster: False = all-pairs filter; True / "group" = two-level filter (k-d clusters);
    # "group-morton" = same with a Morton face order
    order = None
    if cluster:
        order = ops.morton_order(fv) if cluster == "group-morton" else ops.cluster_order(fv)
    sc = ops.Scene3DArgs(fv, cat, mat_in=mat_in, mat_out=mat_out, n_table=n_table,
                         face_grad_mask=gmask, cluster_order=order)
    src = tt(scene["rays"], dtype)
    return src, fv, sc, (p_f, p_b)


def _compare_sets(gpu, gpu_id, ref, tol, what):
    ref_id = ref["ray_id"].numpy().astype(np.int64)
    gid = gpu_id.cpu().numpy().astype(np.int64)
    assert gid.shape == ref_id.shape, f"{what}: {gid.shape[0]} rays vs oracle {ref_id.shape[0]}"
    mism = int((gid != ref_id).sum())
    assert mism == 0, f"{what}: {mism} rays ordered/classified differently"
    g = gpu.detach().cpu().double().numpy()
    r = oracle_util.block(ref)
    scale = max(1.0, np.abs(r).max()) if r.size else 1.0
    err = np.abs(g - r).max() / scale if r.size else 0.0
    assert err <= tol, f"{what}: max rel err {err:.3e} > {tol}"
    return err


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-5)])
def test_forward_lens(dtype, tol):
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(3000, k_front=5, k_back=4)
    src, fv, sc, _ = _gpu_scene(scene, dtype)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    out = ops.trace3d(src, fv, sc, max_passes=5, flags=flags)
    system, _, _ = oracle_util.lens_oracle(scene)
    ref = tracer.ray_trace(
        system, oracle_util.source_dict(scene["rays"], scene["wavelength"],
                                        np.float32 if dtype == torch.float32 else None),
        max_iterations=5, inherit=("wavelength", "ray_id"),
        flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    assert out["finished"].shape[1] > 2000
    for cls in ("finished", "active", "dead"):
        if not ref[cls]:
            assert out[cls].shape[1] == 0
            continue
        _compare_sets(out[cls], out[cls + "_id"], ref[cls], tol, cls)
    M = fv.shape[0]
    assert out["n_tests"] == int(out["counts"][:, :4].sum(axis=1) @ np.ones(5)) * M


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 1e-5)])
def test_backward_lens(dtype, tol):
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(1500, k_front=4, k_back=3)
    rng = np.random.default_rng(5)
    mask_f = (rng.uniform(size=scene["faces_f"].shape) > 0.3)
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype, masks=(mask_f.astype(np.uint8), None))
    out = ops.trace3d(src, fv, sc, max_passes=4)
    fin = out["finished"]
    goal = torch.tensor(scene["goal"], dtype=torch.float64, device=fin.device)[out["finished_id"].long()]
    err = ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
    err = err + 0.1 * (out["active"][3].double() ** 2).sum()  # gradient through the active history too
    g_f, g_b = torch.autograd.grad(err, [p_f, p_b])

    system, (q_f, q_b), _ = oracle_util.lens_oracle(scene, update_map_f=mask_f)
    ref = tracer.ray_trace(
        system, oracle_util.source_dict(scene["rays"], scene["wavelength"],
                                        np.float32 if dtype == torch.float32 else None),
        max_iterations=4, inherit=("wavelength", "ray_id"))
    rf = ref["finished"]
    rgoal = torch.tensor(scene["goal"], dtype=torch.float64)[rf["ray_id"].long()]
    rerr = ((rf["y_end"] - rgoal[:, 0]) ** 2 + (rf["z_end"] - rgoal[:, 1]) ** 2).sum()
    rerr = rerr + 0.1 * (ref["active"]["x_end"] ** 2).sum()
    r_f, r_b = torch.autograd.grad(rerr, [q_f, q_b])
    assert abs(err.item() - rerr.item()) <= tol * max(1.0, abs(rerr.item())) * 10
    for g, r, name in ((g_f, r_f, "front"), (g_b, r_b, "back")):
        g = g.cpu().numpy()
        r = r.numpy()
        rel = np.abs(g - r).max() / np.abs(r).max()
        assert rel <= tol, f"{name} parameter gradient rel err {rel:.3e} > {tol}"


def test_intersection_seam():
    """S1: OpticalSystem3D._intersection (engine.py:1103-1166)."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(2000, k_front=6, k_back=6, seed=3)
    src, fv, sc, _ = _gpu_scene(scene, torch.float64)
    x, y, z, valid, ray_u, trig_u, trig_v, gather = ops.intersect3d(src, fv)
    system, _, _ = oracle_util.lens_oracle(scene)
    m = system.merged
    r = scene["rays"]
    ref = tracer.intersection_3d(*[torch.tensor(r[i]) for i in range(6)],
                                 m["xp"], m["yp"], m["zp"], m["x1"], m["y1"], m["z1"],
                                 m["x2"], m["y2"], m["z2"], 1e-10, 1e-10, 1e-10)
    rv = ref[3].numpy()
    assert np.array_equal(valid.cpu().numpy(), rv)
    assert np.array_equal(gather.cpu().numpy()[rv], ref[8].numpy()[rv])
    for got, want in zip((x, y, z, ray_u, trig_u, trig_v), (ref[0], ref[1], ref[2], ref[4], ref[5], ref[6])):
        np.testing.assert_allclose(got.cpu().numpy()[rv], want.detach().numpy()[rv], rtol=1e-12, atol=1e-12)


def test_float16_ray_state_experiment():
    """BASELINE config 5: fp16 ray *storage* is an accuracy experiment, not held to 1e-5."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(3000, k_front=5, k_back=4)
    src, fv, sc, _ = _gpu_scene(scene, torch.float16)
    out = ops.trace3d(src, fv, sc, max_passes=4)
    # the same float16 state through the default (grouped) kernel: identical to all-pairs
    srcg, fvg, scg, _ = _gpu_scene(scene, torch.float16, cluster="group")
    outg = ops.trace3d(srcg, fvg, scg, max_passes=4)
    assert torch.equal(outg["finished"], out["finished"])
    assert torch.equal(outg["finished_id"], out["finished_id"])
    src64, fv64, sc64, _ = _gpu_scene(scene, torch.float64)
    ref = ops.trace3d(src64, fv64, sc64, max_passes=4)
    n16, n64 = out["finished"].shape[1], ref["finished"].shape[1]
    assert abs(n16 - n64) <= 0.02 * n64
    both = np.intersect1d(out["finished_id"].cpu().numpy(), ref["finished_id"].cpu().numpy())
    assert both.size > 0.95 * n64
    def by_id(o):
        ids = o["finished_id"].cpu().numpy()
        keep = np.isin(ids, both)
        order = np.argsort(ids[keep])
        return o["finished"].detach().double().cpu().numpy()[:, keep][:, order]

    err = np.abs(by_id(out) - by_id(ref))
    # half precision: ~1e-3 direction error carried over the 10-unit throw to the target
    assert np.median(err) < 2e-2 and np.quantile(err, 0.99) < 0.5


def test_value_mode_matches_index_mode():
    """StandardReaction('value'): per-face n_in / n_out instead of material indices."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(1500, k_front=3, k_back=3)
    src, fv, sc, _ = _gpu_scene(scene, torch.float64)
    a = ops.trace3d(src, fv, sc, max_passes=4)
    n_acr = float(tracer.MATERIALS["acrylic"](torch.tensor([575.0], dtype=torch.float64)))
    M = fv.shape[0]
    n_in = torch.where(sc.catagory == 0, torch.full((M,), n_acr, dtype=torch.float64, device=fv.device),
                       torch.ones(M, dtype=torch.float64, device=fv.device))
    n_out = torch.ones(M, dtype=torch.float64, device=fv.device)
    sv = ops.Scene3DArgs(fv, sc.catagory, n_in=n_in, n_out=n_out)
    b = ops.trace3d(src, fv, sv, max_passes=4)
    assert torch.equal(a["finished_id"], b["finished_id"])
    np.testing.assert_allclose(a["finished"].detach().cpu().numpy(), b["finished"].detach().cpu().numpy(), atol=1e-12)


@pytest.mark.parametrize("coherent", [False, True])
def test_value_mode_gradient_with_respect_to_the_refractive_indices(coherent):
    """StandardReaction('value') reads n_in / n_out as ordinary tensors (operation.py:268-272), so
    a tape can differentiate an error w.r.t. them: the reverse sweep accumulates d error /
    d n_in[face], d n_out[face] (tfrt_scene3d.grad_n_in / grad_n_out) -- against torch.autograd
    through the oracle, per-face indices drawn at random around the acrylic value."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(6000, k_front=4, k_back=3)
    src, fv, sc, _ = _gpu_scene(scene, torch.float64, cluster="group")
    M = fv.shape[0]
    rng = np.random.default_rng(9)
    optical = (sc.catagory == 0).cpu().numpy()
    n_in_np = np.where(optical, 1.49 + 0.05 * rng.standard_normal(M), 1.0)
    n_out_np = np.where(optical, 1.0 + 0.02 * rng.random(M), 1.0)
    n_in = torch.tensor(n_in_np, device=fv.device, requires_grad=True)
    n_out = torch.tensor(n_out_np, device=fv.device, requires_grad=True)
    sv = ops.Scene3DArgs(fv.detach(), sc.catagory, n_in=n_in, n_out=n_out,
                         cluster_order=sc.cluster_order, coherent_rays=coherent)
    out = ops.trace3d(src, fv.detach(), sv, max_passes=4)
    goal = torch.tensor(scene["goal"], dtype=torch.float64, device=fv.device)[out["finished_id"].long()]
    fin = out["finished"]
    loss = ((fin[4] - goal[:, 0]) ** 2 + (fin[5] - goal[:, 1]) ** 2).sum()
    g_in, g_out = torch.autograd.grad(loss, [n_in, n_out])

    system, _, fields = oracle_util.lens_oracle(scene)
    nf = fields["front"]["xp"].shape[0] + fields["back"]["xp"].shape[0]
    o_in = torch.tensor(n_in_np[:nf], requires_grad=True)
    o_out = torch.tensor(n_out_np[:nf], requires_grad=True)
    system.optical["n_in"], system.optical["n_out"] = o_in, o_out
    ref = tracer.ray_trace(system, oracle_util.source_dict(scene["rays"], scene["wavelength"]),
                           max_iterations=4, inherit=("wavelength", "ray_id"), index_type="value")
    rf = ref["finished"]
    assert np.array_equal(out["finished_id"].cpu().numpy(), rf["ray_id"].numpy().astype(np.int32))
    rgoal = torch.tensor(scene["goal"], dtype=torch.float64)[rf["ray_id"].long()]
    rloss = ((rf["y_end"] - rgoal[:, 0]) ** 2 + (rf["z_end"] - rgoal[:, 1]) ** 2).sum()
    r_in, r_out = torch.autograd.grad(rloss, [o_in, o_out])
    assert abs(loss.item() - rloss.item()) <= 1e-9 * rloss.item()
    for got, want in ((g_in, r_in), (g_out, r_out)):
        assert float(want.abs().max()) > 0
        rel = float((got.cpu()[:nf] - want).abs().max() / want.abs().max())
        assert rel < 1e-8, rel
        assert float(got[nf:].abs().max()) == 0.0          # the target plane refracts nothing


def test_gradients_at_the_critical_angle_and_float32_against_float64_state():
    """(i) Rays inside acrylic meeting a flat face within 1e-9 rad of the critical angle, on both
    sides: the reverse sweep takes the branch the forward pass took (recorded on the tape: two bits
    of the class byte), so the parameter gradient is finite and refracting / reflecting rays get
    the gradient of their own branch (oracle autograd).  (ii) The float32-state sweep (ray
    gradients between passes stored in float32) stays within 5e-6 of the float64-state one."""
    from tensorflowraytrace_amd import ops, _lib
    dev = "cuda:0"
    n_glass = 1.5
    crit = np.arcsin(1.0 / n_glass)
    # one big face in the plane x = 0 (norm +x: n_in = glass on the -x side), a target far away
    P = torch.tensor([[0.0, -50.0, -50.0, 0.0, 50.0, -50.0, 0.0, 0.0, 80.0],
                      [40.0, -500.0, -500.0, 40.0, 500.0, -500.0, 40.0, 0.0, 800.0],
                      [-40.0, -500.0, -500.0, -40.0, 0.0, 800.0, -40.0, 500.0, -500.0],
                      # (a wall across the surface: rays refracted to grazing exit end here)
                      [-100.0, 30.0, -300.0, 100.0, 30.0, -300.0, 0.0, 30.0, 600.0]],
                     dtype=torch.float64, device=dev, requires_grad=True)
    cat = torch.tensor([0, 2, 2, 2], dtype=torch.int32, device=dev)
    off = np.array([-1e-9, -1e-12, 1e-12, 1e-9, -0.2, 0.1])     # around the critical angle
    ang = crit + off
    s = np.stack([-np.cos(ang), -np.sin(ang), np.zeros_like(ang)])      # from inside the glass
    e = np.zeros_like(s)
    e[2] = 0.01
    rays = torch.tensor(np.concatenate([s, e + 0.0 * s]), dtype=torch.float64, device=dev)
    rays[3:] = rays[:3] * 0.5                                            # ends half-way to the face
    args = ops.Scene3DArgs(P.detach(), cat, n_in=torch.tensor([n_glass, 1.0, 1.0, 1.0], device=dev),
                           n_out=torch.ones(4, dtype=torch.float64, device=dev))
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    out = ops.trace3d(rays, P, args, max_passes=2, flags=flags)
    assert out["finished"].shape[1] == 6                                 # refracted out or reflected back
    loss = (out["finished"][4] ** 2).sum() + out["finished"][5].sum()
    (g,) = torch.autograd.grad(loss, [P])
    assert bool(torch.isfinite(g).all())
    from oracle import tracer as otr
    Pc = P.detach().cpu().clone().requires_grad_(True)
    faces = otr.faces_from_vertices(Pc.reshape(-1, 3), torch.arange(12).reshape(4, 3))
    sub = lambda m: {k: v[m] for k, v in faces.items()}
    opt_f = sub(torch.tensor([True, False, False, False]))
    opt_f["n_in"], opt_f["n_out"] = torch.tensor([n_glass]), torch.tensor([1.0])
    system = otr.System(3, optical=opt_f, target=sub(torch.tensor([False, True, True, True])))
    src = {k: rays[i].cpu() for i, k in enumerate(("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"))}
    src["ray_id"] = torch.arange(6, dtype=torch.float64)
    ref = otr.ray_trace(system, src, max_iterations=2, inherit=("ray_id",), index_type="value")
    assert np.array_equal(out["finished_id"].cpu().numpy(), ref["finished"]["ray_id"].numpy().astype(np.int32))
    rloss = (ref["finished"]["y_end"] ** 2).sum() + ref["finished"]["z_end"].sum()
    (rg,) = torch.autograd.grad(rloss, [Pc])
    assert float((g.cpu() - rg).abs().max() / rg.abs().max()) < 1e-7

    # (ii) float32 against float64 ray state, lens scene, three passes
    scene = scene_util.lens_scene(20000, k_front=8, k_back=6)
    grads = {}
    for dt in (torch.float32, torch.float64):
        src_b, fv, sc, (p_f, p_b) = _gpu_scene(scene, dt, cluster="group")
        o = ops.trace3d(src_b, fv, sc, max_passes=4)
        goal = torch.tensor(scene["goal"], dtype=torch.float64, device=dev)[o["finished_id"].long()]
        err = ((o["finished"][4].double() - goal[:, 0]) ** 2 + (o["finished"][5].double() - goal[:, 1]) ** 2).sum()
        grads[dt] = torch.autograd.grad(err, [p_f, p_b])
    for a, b in zip(grads[torch.float32], grads[torch.float64]):
        assert float((a - b).abs().max() / b.abs().max()) < 5e-6


def test_empty_and_degenerate_inputs():
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(64, k_front=2, k_back=2)
    src, fv, sc, _ = _gpu_scene(scene, torch.float32)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    # no rays
    out = ops.trace3d(src[:, :0].contiguous(), fv, sc, max_passes=3, flags=flags)
    assert out["finished"].shape == (6, 0) and out["n_tests"] == 0
    # no faces: everything dies in pass 0
    dev = src.device
    empty = ops.Scene3DArgs(fv[:0], sc.catagory[:0], mat_in=sc.mat_in[:0], mat_out=sc.mat_out[:0],
                            n_table=sc.n_table)
    out = ops.trace3d(src, fv[:0].detach(), empty, max_passes=3, flags=flags)
    assert out["dead"].shape[1] == 64 and out["finished"].shape[1] == 0
    # zero-length and NaN-free handling: a zero-length ray never hits (den == 0 in the reference)
    s2 = src.clone()
    s2[3:, :5] = s2[:3, :5]
    out = ops.trace3d(s2, fv, sc, max_passes=3, flags=flags)
    dead_ids = set(out["dead_id"].cpu().numpy().tolist())
    assert {0, 1, 2, 3, 4} <= dead_ids
    # a single ray / a single face
    out = ops.trace3d(src[:, :1].contiguous(), fv, sc, max_passes=3, flags=flags)
    assert out["finished"].shape[1] + out["dead"].shape[1] + out["unfinished"].shape[1] == 1


@pytest.mark.parametrize("mode", ["group", "group-morton"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-5)])
def test_clustered_path_matches_oracle_and_all_pairs(dtype, tol, mode):
    """cluster_order given (sphere hierarchy): results identical
    to the all-pairs filter, bit for bit."""
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(5000, k_front=8, k_back=6)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype, cluster=mode)
    assert fv.shape[0] >= 64
    out = ops.trace3d(src, fv, sc, max_passes=5, flags=flags)
    src2, fv2, sc2, (q_f, q_b) = _gpu_scene(scene, dtype, cluster=False)
    ref = ops.trace3d(src2, fv2, sc2, max_passes=5, flags=flags)
    assert np.array_equal(out["counts"], ref["counts"]) and out["n_tests"] == ref["n_tests"]
    for cls in ("finished", "active", "dead"):
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"])
        assert torch.equal(out[cls + "_face"], ref[cls + "_face"])
        assert torch.equal(out[cls], ref[cls])          # bit-identical ray blocks
    # against the oracle
    system, _, _ = oracle_util.lens_oracle(scene)
    oref = tracer.ray_trace(
        system, oracle_util.source_dict(scene["rays"], scene["wavelength"],
                                        np.float32 if dtype == torch.float32 else None),
        max_iterations=5, inherit=("wavelength", "ray_id"), flags=dict(compile_dead_rays=True))
    _compare_sets(out["finished"], out["finished_id"], oref["finished"], tol, "finished")
    # gradients: same up to the order of the float64 sums
    def grads(o, params):
        fin = o["finished"]
        goal = torch.tensor(scene["goal"], dtype=torch.float64, device=fin.device)[o["finished_id"].long()]
        err = ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
        return torch.autograd.grad(err, params)
    ga, gb = grads(out, [p_f, p_b]), grads(ref, [q_f, q_b])
    for a, b in zip(ga, gb):
        rel = float((a - b).abs().max() / b.abs().max())
        assert rel < 1e-11, rel


@pytest.mark.parametrize("n_rays", [3000, 70000])
def test_ties_go_to_the_lowest_face_index_in_every_trace_mode(n_rays):
    """tf.argmin picks the first index among equal ray_u (engine.py:1148).  Every face of the
    scene appears twice (index j and j + M): each hit is an exact tie, and all three kernels
    must report the lower copy although the clustered ones visit faces out of index order.
    One pass only: a ray leaving face j would immediately hit its twin j + M."""
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(n_rays, k_front=8, k_back=6)
    src, fv, sc, _ = _gpu_scene(scene, torch.float32)
    M = fv.shape[0]
    fv2 = torch.cat([fv, fv]).detach()
    dup = lambda t: torch.cat([t, t])
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD
    single = ops.trace3d(src, fv.detach(), sc, max_passes=1, flags=flags)
    assert single["active"].shape[1] > 0.8 * n_rays
    for mode in (False, "group"):
        order = ops.cluster_order(fv2) if mode else None
        sc2 = ops.Scene3DArgs(fv2, dup(sc.catagory), mat_in=dup(sc.mat_in), mat_out=dup(sc.mat_out),
                              n_table=sc.n_table, cluster_order=order)
        o = ops.trace3d(src, fv2, sc2, max_passes=1, flags=flags)
        assert int(o["active_face"].max()) < M, mode
        for cls in ("active", "dead"):
            assert torch.equal(o[cls + "_face"], single[cls + "_face"]), mode
            assert torch.equal(o[cls], single[cls]), mode
        assert torch.equal(o["unfinished"], single["unfinished"]), mode


def _soup_scene(seed, n_faces, n_rays, dev="cuda:0"):
    """Random triangle soup (all sizes, some degenerate / duplicated faces, one stop and one
    target region) and random rays: nothing mesh-like, to stress the conservative filters."""
    from tensorflowraytrace_amd import ops
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-1, 1, (n_faces, 1, 3))
    size = 10 ** rng.uniform(-2.5, -0.3, (n_faces, 1, 1))
    tri = centre + size * rng.standard_normal((n_faces, 3, 3))
    tri[::37, 2] = tri[::37, 1]                         # zero-area faces
    if n_faces > 50:
        tri[40:45] = tri[10:15]                         # exact duplicates (ties)
    fv = torch.tensor(tri.reshape(n_faces, 9), dtype=torch.float64, device=dev)
    # merged order of the engine (engine.py:971-1018): optical, then stops, then targets
    cat = torch.zeros(n_faces, dtype=torch.int32, device=dev)
    cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1
    cat[int(0.9 * n_faces):] = 2
    n_in = torch.tensor(rng.uniform(1.0, 1.7, n_faces), device=dev)
    n_out = torch.tensor(rng.uniform(1.0, 1.7, n_faces), device=dev)
    s = rng.uniform(-1.5, 1.5, (3, n_rays))
    e = s + rng.standard_normal((3, n_rays)) * 0.7
    rays = torch.tensor(np.concatenate([s, e]), dtype=torch.float32, device=dev)

    def scene(mode):
        order = None
        if mode:
            order = ops.morton_order(fv) if mode == "group-morton" else ops.cluster_order(fv)
        return ops.Scene3DArgs(fv, cat, n_in=n_in, n_out=n_out, cluster_order=order)
    return rays, fv, scene


@pytest.mark.parametrize("seed,n_faces,n_rays", [
    (1, 64, 1), (2, 65, 63), (3, 100, 257), (4, 129, 5000), (5, 1000, 20000), (6, 4097, 3000),
    (7, 30000, 700)])
def test_every_trace_mode_gives_the_all_pairs_result_on_random_soups(seed, n_faces, n_rays):
    """The sphere hierarchy, and the float32 screen may only skip work that
    cannot change the result: all outputs must equal the all-pairs filter's bit for bit."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(seed, n_faces, n_rays)
    flags = (_lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED)
    ref = ops.trace3d(rays, fv, scene(False), max_passes=4, flags=flags, dead_ray_length=2.0)
    assert ref["active"].shape[1] > 0 or n_rays < 10
    for mode in ("group", "group-morton"):
        out = ops.trace3d(rays, fv, scene(mode), max_passes=4, flags=flags, dead_ray_length=2.0)
        assert np.array_equal(out["counts"], ref["counts"]), mode
        for cls in ("finished", "active", "stopped", "dead"):
            assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (mode, cls)
            assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), (mode, cls)
            assert torch.equal(out[cls], ref[cls]), (mode, cls)
        assert torch.equal(out["unfinished"], ref["unfinished"]), mode


def test_soup_matches_oracle_in_default_mode():
    """The same kind of soup against the float64 restatement of the reference (value mode)."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(11, 300, 4000)
    rays = rays.double()
    sc = scene("group")
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    out = ops.trace3d(rays, fv, sc, max_passes=3, flags=flags)
    cat = sc.catagory.cpu().long()

    def sub(mask):
        verts = fv.cpu()[mask].reshape(-1, 3)
        d = tracer.faces_from_vertices(verts, torch.arange(verts.shape[0]).reshape(-1, 3))
        d["n_in"] = sc.n_in.cpu()[mask]
        d["n_out"] = sc.n_out.cpu()[mask]
        return d

    system = tracer.System(3, optical=sub(cat == 0), stop=sub(cat == 1), target=sub(cat == 2))
    r = rays.cpu()
    src = {n: r[i] for i, n in enumerate(("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"))}
    src["ray_id"] = torch.arange(r.shape[1])
    ref = tracer.ray_trace(system, src, max_iterations=3, inherit=("ray_id",), index_type="value",
                           flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    for cls in ("finished", "active", "stopped", "dead"):
        if not ref[cls]:
            assert out[cls].shape[1] == 0
            continue
        _compare_sets(out[cls], out[cls + "_id"], ref[cls], 1e-9, cls)


def test_grouped_kernel_with_several_cluster_chunks():
    """A small launch spreads the scene over several cluster chunks (here 36 ray blocks x 2
    chunks of the 125 clusters; k_classify3d then merges the partial results): the all-pairs
    result bit for bit."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(21, 2000, 9000)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    ref = ops.trace3d(rays, fv, scene(False), max_passes=3, flags=flags)
    out = ops.trace3d(rays, fv, scene("group"), max_passes=3, flags=flags)
    assert np.array_equal(out["counts"], ref["counts"])
    for cls in ("finished", "active", "stopped", "dead"):
        assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), cls
        assert torch.equal(out[cls], ref[cls]), cls


@pytest.mark.parametrize("n_rays", [1, 255, 257, 9001])
def test_grouped_kernel_classifies_in_its_epilogue_when_it_runs_as_one_chunk(n_rays):
    """With a single cluster chunk (big traces; here a scene of 63 clusters, which is never
    split) k_intersect_group writes the hit records, classes and block histograms itself instead
    of k_classify3d: counts, order and rays must equal the all-pairs path, also for ray counts
    that leave the last 256-ray slice partly or wholly empty."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(77, 1000, 9001)
    rays = rays[:, :n_rays].contiguous()
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    ref = ops.trace3d(rays, fv, scene(False), max_passes=3, flags=flags)
    out = ops.trace3d(rays, fv, scene("group"), max_passes=3, flags=flags)
    assert np.array_equal(out["counts"], ref["counts"])
    assert out["n_tests"] == ref["n_tests"]
    for cls in ("finished", "active", "stopped", "dead"):
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), cls
        assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), cls
        assert torch.equal(out[cls], ref[cls]), cls


@pytest.mark.parametrize("mode", ["group"])
def test_empty_and_tiny_inputs_in_the_cluster_modes(mode):
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(31, 200, 300)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    sc = scene(mode)
    out = ops.trace3d(rays[:, :0].contiguous(), fv, sc, max_passes=3, flags=flags)
    assert all(out[c].shape[1] == 0 for c in ("finished", "active", "stopped", "dead"))
    assert out["n_tests"] == 0
    one = ops.trace3d(rays[:, :1].contiguous(), fv, sc, max_passes=3, flags=flags)
    ref = ops.trace3d(rays[:, :1].contiguous(), fv, scene(False), max_passes=3, flags=flags)
    for cls in ("finished", "active", "stopped", "dead"):
        assert torch.equal(one[cls], ref[cls])
    # rays that miss everything: all dead after one pass
    far = rays.clone()
    far[:3] += 100.0
    far[3:] += 100.0
    far[5] += 1.0
    miss = ops.trace3d(far, fv, sc, max_passes=3, flags=flags)
    assert miss["dead"].shape[1] == far.shape[1] and miss["active"].shape[1] == 0


def test_lanes_without_rays_never_queue_candidates():
    """Regression: size_epsilion = 1e300 inflates every bounding sphere to +inf; lanes without
    a ray used +inf offsets, passed `inf <= inf`, queued candidates and read ray slots >= n
    (an intermittent out-of-bounds fault in the 1-ray seam calls of the golden test).  With
    a handful of rays in a 256-lane workgroup the results must equal the plain evaluation."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(41, 200, 5)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    for mode in (False, "group"):
        sc = scene(mode)
        sc.eps = (1e-10, 1e300, 1e-10)           # size_epsilion: every face becomes "all space"
        out = ops.trace3d(rays, fv, sc, max_passes=2, flags=flags)
        total = sum(out[c].shape[1] for c in ("finished", "stopped", "dead")) + out["unfinished"].shape[1]
        assert total == 5, mode
    x, y, z, valid, ray_u, tu, tv, gi = ops.intersect3d(rays.double(), fv, 1e-10, 1e300, -1e300)
    assert valid.shape[0] == 5 and bool(valid.all())    # with these epsilons every plane is hit


def test_negative_ray_start_epsilon_keeps_hits_just_behind_the_start():
    """The grouped filter drops clusters behind a ray's start from the second pass on -- only
    while ray_start_epsilion >= 0.  With a negative epsilon the reference accepts hits at
    slightly negative ray_u (the face a ray has just left is excluded by index, its neighbours in
    the same plane are not): every trace mode must still agree with the all-pairs filter."""
    from tensorflowraytrace_amd import ops, _lib
    rays, fv, scene = _soup_scene(5, 900, 6000)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED

    def run(mode):
        sc = scene(mode)
        sc.eps = (sc.eps[0], sc.eps[1], -0.05)          # ray_start_epsilion
        sc._struct_cache = None
        return ops.trace3d(rays, fv, sc, max_passes=3, flags=flags)

    ref = run(False)
    out = run("group")
    assert np.array_equal(out["counts"], ref["counts"])
    for cls in ("finished", "active", "stopped", "dead"):
        assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), cls
        assert torch.equal(out[cls], ref[cls]), cls


def test_non_finite_ray_gradient_poisons_the_same_entries_as_in_the_oracle():
    """optimizer.py:226-229 zeroes non-finite entries of the SUMMED parameter gradient: one ray
    with a NaN gradient must therefore poison every parameter it touches (and only those) -- the
    reverse sweep lets NaN through its sums instead of dropping the ray.  A NaN weight on one
    finished ray's error term stands in for a degenerate ray."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(600, k_front=3, k_back=3)
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, torch.float64)
    out = ops.trace3d(src, fv, sc, max_passes=4)
    fin, ids = out["finished"], out["finished_id"].long()
    assert fin.shape[1] > 400
    weight = torch.ones(600, dtype=torch.float64)
    # (the LAST finished ray: it went through both lens surfaces; the first rows of the finished
    # set are rays that pass outside the hexagonal lens and touch no parameter at all)
    weight[ids[-1].item()] = float("nan")
    loss = (weight.to(fin.device)[ids] * (fin[4] ** 2 + fin[5] ** 2)).sum()
    g_f, g_b = torch.autograd.grad(loss, [p_f, p_b])

    system, (q_f, q_b), _ = oracle_util.lens_oracle(scene)
    ref = tracer.ray_trace(system, oracle_util.source_dict(scene["rays"], scene["wavelength"]),
                           max_iterations=4, inherit=("wavelength", "ray_id"))
    rf = ref["finished"]
    r_loss = (weight[rf["ray_id"].long()] * (rf["y_end"] ** 2 + rf["z_end"] ** 2)).sum()
    r_f, r_b = torch.autograd.grad(r_loss, [q_f, q_b])
    for g, r in ((g_f.cpu(), r_f), (g_b.cpu(), r_b)):
        bad = ~torch.isfinite(r)
        assert 0 < int(bad.sum()) <= 6                         # the vertices of the faces that ray hit
        assert torch.equal(~torch.isfinite(g), bad)
        assert float((g[~bad] - r[~bad]).abs().max()) <= 1e-8 * float(r[~bad].abs().max())
    # the optimiser's processing then zeroes exactly those entries (tfrt_sgd_process)
    done = ops.sgd_process(g_f, 1.0, 1e30)
    assert bool(torch.isfinite(done).all()) and bool((done.cpu()[~torch.isfinite(r_f)] == 0).all())


def test_ordered_reverse_sweep_is_bit_reproducible_and_agrees_with_the_atomic_one():
    """tfrt_scene3d.deterministic: face gradients summed as scaled 64-bit integers.  Three runs
    give identical bits (parameter gradients included: the face -> vertex reverse is a gather in a
    fixed corner order), and the result equals the default float64-atomic sweep to 1e-10 of the
    largest gradient entry."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(300_000, k_front=12, k_back=6)

    def grads(deterministic):
        src, fv, sc, (p_f, p_b) = _gpu_scene(scene, torch.float32, cluster="group")
        sc.deterministic = deterministic
        out = ops.trace3d(src, fv, sc, max_passes=3)
        fin = out["finished"]
        goal = torch.tensor(scene["goal"], dtype=torch.float64, device=fin.device)[out["finished_id"].long()]
        loss = ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
        return [g.clone() for g in torch.autograd.grad(loss, [p_f, p_b])]

    runs = [grads(True) for _ in range(3)]
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    ref = grads(False)
    for a, b in zip(runs[0], ref):
        assert float((a - b).abs().max()) <= 1e-10 * float(b.abs().max())
        assert float(b.abs().max()) > 0


# --------------------------------------------------------------------------------------------
# Coherent rays (tfrt_scene3d.coherent_rays): the caller hands the rays over sorted so that
# neighbours have neighbouring lines; wavefronts then share one walk of the hierarchy
# (k_intersect_beam), the others take the per-ray walk (k_intersect_group), and
# ops.restore_order() gives back the ray sets of the unsorted rays -- nothing else may change.

def _orders(src):
    from tensorflowraytrace_amd import ops
    n = src.shape[1]
    g = torch.Generator(device="cpu").manual_seed(7)
    return {
        "hilbert": ops.ray_order(src),                                   # coherent: the beam kernel
        "random": torch.randperm(n, generator=g).int().to(src.device),   # not coherent: the grouped kernel
        "identity": torch.arange(n, dtype=torch.int32, device=src.device),
    }


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n_rays,k_front", [(20000, 12), (70000, 20)])
def test_coherent_order_changes_no_output(dtype, n_rays, k_front):
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(n_rays, k_front=k_front, k_back=6)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype, cluster="group")
    ref = ops.trace3d(src, fv, sc, max_passes=4, flags=flags)

    def loss(o):
        fin = o["finished"]
        goal = torch.tensor(scene["goal"], dtype=torch.float64, device=fin.device)[o["finished_id"].long()]
        return ((fin[4].double() - goal[:, 0]) ** 2 + (fin[5].double() - goal[:, 1]) ** 2).sum()
    g_ref = torch.autograd.grad(loss(ref), [p_f, p_b], retain_graph=True)
    runs = [(name, order, False) for name, order in _orders(src).items()]
    # coherent_only: no grouped-kernel launch behind k_intersect_beam, which then finishes every
    # wavefront itself -- also the ones that are not coherent (cut down to single rays)
    runs += [(name + "+only", order, True) for name, order in _orders(src).items()
             if name != "identity"]
    for name, order, only in runs:
        p64 = order.long()
        sc2 = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                              n_table=sc.n_table[:, p64].contiguous(),
                              face_grad_mask=sc.face_grad_mask, cluster_order=sc.cluster_order,
                              coherent_rays=True)
        sc2.coherent_only = only
        raw = ops.trace3d(src[:, p64].contiguous(), fv, sc2, max_passes=4, flags=flags)
        if name == "hilbert":
            assert raw["left_over"] == 0           # coherent: nothing for the grouped kernel
        if name == "random":
            assert raw["left_over"] > 0
        out = ops.restore_order(raw, order)
        assert np.array_equal(out["counts"], ref["counts"]), name
        assert out["n_tests"] == ref["n_tests"], name
        for cls in ("finished", "active", "dead", "stopped", "unfinished"):
            assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (name, cls)
            if cls != "unfinished":
                assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), (name, cls)
            assert torch.equal(out[cls], ref[cls]), (name, cls)          # every bit
        g = torch.autograd.grad(loss(out), [p_f, p_b], retain_graph=True)
        # float64 state: the same terms summed in another order.  float32 state: the natural-order
        # sweep rounds every ray's nine face terms to float32 on their way through the stash, the
        # coherent sweep sums them in float64 straight away (the float32 level: 6e-8)
        tol = 1e-11 if dtype == torch.float64 else 2e-6
        for a, b in zip(g, g_ref):
            assert float((a - b).abs().max() / b.abs().max()) < tol, name


@pytest.mark.parametrize("max_passes", [1, 8, 10])
def test_coherent_reverse_sweep_at_every_trace_depth(max_passes):
    """The one-launch reverse sweep (k_backward_chain) holds a ray's chain of slots for up to eight
    passes; deeper traces take the per-pass sweep.  Either way, and for a one-pass trace (the
    chain is the source slot alone), the gradients are those of the natural-order trace."""
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(20000, k_front=12, k_back=6)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, torch.float64, cluster="group")

    def loss(o):      # every class and both ends of every row carry a gradient
        tot = 0.0
        for k, cls in enumerate(("finished", "active", "stopped", "dead")):
            r = o[cls].double()
            w = torch.arange(1, 7, dtype=torch.float64, device=r.device)[:, None] * (0.3 + k)
            tot = tot + (w * r * r).sum() + (r[3:] * r[:3]).sum()
        return tot
    ref = ops.trace3d(src, fv, sc, max_passes=max_passes, flags=flags)
    g_ref = torch.autograd.grad(loss(ref), [p_f, p_b], retain_graph=True)
    order = ops.ray_order(src)
    p64 = order.long()
    sc2 = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                          n_table=sc.n_table[:, p64].contiguous(), face_grad_mask=sc.face_grad_mask,
                          cluster_order=sc.cluster_order, coherent_rays=True)
    raw = ops.trace3d(src[:, p64].contiguous(), fv, sc2, max_passes=max_passes, flags=flags)
    out = ops.restore_order(raw, order)
    assert np.array_equal(out["counts"], ref["counts"])
    g = torch.autograd.grad(loss(out), [p_f, p_b], retain_graph=True)
    assert float(g_ref[0].abs().max()) > 0.0
    for a, b in zip(g, g_ref):
        if float(b.abs().max()) == 0.0:      # (one pass: the back surface is never reached)
            assert float(a.abs().max()) == 0.0
        else:
            assert float((a - b).abs().max() / b.abs().max()) < 1e-11


def test_coherent_reverse_sweep_gives_the_source_ray_gradient():
    """grad_src_rays out of the one-launch reverse sweep (a lane ends its walk back at its source
    ray): equal to the per-pass sweep's, for rays handed over in a coherent order."""
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(20000, k_front=12, k_back=6)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, torch.float64, cluster="group")
    order = ops.ray_order(src).long()
    grads = {}
    for coherent in (False, True):
        rays = src[:, order].contiguous().requires_grad_(True)
        sc2 = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                              n_table=sc.n_table[:, order].contiguous(),
                              face_grad_mask=sc.face_grad_mask, cluster_order=sc.cluster_order,
                              coherent_rays=coherent)
        out = ops.trace3d(rays, fv, sc2, max_passes=4, flags=flags)
        loss = sum((o.double() ** 2).sum() * (k + 1) for k, o in
                   enumerate(out[c] for c in ("finished", "active", "stopped", "dead")))
        grads[coherent] = torch.autograd.grad(loss, [rays, p_f, p_b], retain_graph=True)
    for a, b in zip(grads[True], grads[False]):
        assert float(b.abs().max()) > 0.0
        assert float((a - b).abs().max() / b.abs().max()) < 1e-11


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("coherent", [False, True])
def test_one_wavelength_table_and_per_face_tables_change_no_bit(dtype, coherent):
    """tfrt_scene3d.n_table_uniform (one table column read by every ray) lets the trace's set-up
    launch form the face's unit normal AND its index ratios once per face (FaceTables) and the
    reverse sweep read the indices per face: against the same trace with one table column per
    ray -- the reaction's per-ray path -- every output bit is the same, in natural and in
    coherent order, and the gradients agree to the last bits of a differently ordered sum."""
    from tensorflowraytrace_amd import ops, _lib
    scene = scene_util.lens_scene(30000, k_front=12, k_back=6)
    scene["wavelength"] = np.full_like(np.asarray(scene["wavelength"], dtype=np.float64), 587.6)
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype, cluster="group")
    if coherent:
        order = ops.ray_order(src).long()
        src = src[:, order].contiguous()

    def run(uniform):
        table = sc.n_table[:, :1].contiguous() if uniform else sc.n_table
        sc2 = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out, n_table=table,
                              face_grad_mask=sc.face_grad_mask, cluster_order=sc.cluster_order,
                              coherent_rays=coherent)
        sc2.n_table_uniform = uniform
        out = ops.trace3d(src, fv, sc2, max_passes=4, flags=flags)
        loss = sum((out[c].double() ** 2).sum() * (k + 1)
                   for k, c in enumerate(("finished", "active", "stopped", "dead")))
        return out, torch.autograd.grad(loss, [p_f, p_b], retain_graph=True)
    ref, g_ref = run(False)
    out, g = run(True)
    assert np.array_equal(out["counts"], ref["counts"])
    for cls in ("finished", "active", "dead", "stopped", "unfinished"):
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), cls
        assert torch.equal(out[cls], ref[cls]), cls            # every bit
    assert ref["finished"].shape[1] > 1000
    for a, b in zip(g, g_ref):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-11


def test_coherent_flag_on_adversarial_soups():
    """Random triangle soups, random rays (no coherence at all, grazing rays, ties, stops): with
    the flag set and any order of the rays the all-pairs result comes back bit for bit."""
    from tensorflowraytrace_amd import ops, _lib
    import test_gpu_stress as st
    DEV = "cuda:0"
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    for seed in (35, 2, 16, 25, 7):
        sc0 = st._soup(seed)
        fv = sc0["P"].to(DEV)
        rays = sc0["rays"].to(DEV)
        if rays.shape[1] < 64:
            continue
        base = dict(n_in=sc0["n_in"].to(DEV), n_out=sc0["n_out"].to(DEV))
        plain = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), **base)
        ref = ops.trace3d(rays, fv, plain, max_passes=4, flags=flags, new_ray_length=sc0["L"])
        for name, order in _orders(rays).items():
            args = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv),
                                   coherent_rays=True, **base)
            raw = ops.trace3d(rays[:, order.long()].contiguous(), fv, args, max_passes=4, flags=flags,
                              new_ray_length=sc0["L"])
            out = ops.restore_order(raw, order)
            for cls in ("finished", "active", "dead", "stopped", "unfinished"):
                assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (seed, name, cls)
                assert torch.equal(out[cls], ref[cls]), (seed, name, cls)


@pytest.mark.parametrize("eps", [(1e-10, 0.05, -0.05), (1e-10, 1e-3, 1e-6), (1e-6, 0.3, 0.0),
                                 (1e-10, 1e300, 1e-10)])
def test_coherent_order_with_unusual_epsilons(eps):
    """The bounds of k_intersect_beam carry size_epsilion (a valid hit may lie that far outside
    its triangle) and drop the "behind the start" cull when ray_start_epsilion < 0: with large,
    negative or absurd epsilons the sorted trace still returns the natural-order one bit for bit,
    on the lens (coherent wavefronts) and on random soups (cuts, single rays, left-over)."""
    from tensorflowraytrace_amd import ops, _lib
    import test_gpu_stress as st
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    scene = scene_util.lens_scene(30000, k_front=12, k_back=6)
    src, fv, sc, _ = _gpu_scene(scene, torch.float32, cluster="group")
    sc.eps = eps
    ref = ops.trace3d(src, fv.detach(), sc, max_passes=4, flags=flags)
    cases = [(src, fv.detach(), sc, ref, 1.0, "lens")]
    for seed in (35, 16):
        sc0 = st._soup(seed)
        fvs, rays = sc0["P"].to("cuda:0"), sc0["rays"].to("cuda:0")
        if rays.shape[1] < 64:
            continue
        base = dict(n_in=sc0["n_in"].to("cuda:0"), n_out=sc0["n_out"].to("cuda:0"))
        plain = ops.Scene3DArgs(fvs, sc0["cat"].int().to("cuda:0"), **base)
        plain.eps = eps
        cases.append((rays, fvs, plain, ops.trace3d(rays, fvs, plain, max_passes=4, flags=flags,
                                                    new_ray_length=sc0["L"]), sc0["L"], f"soup{seed}"))
    for rays, faces, args0, want, L, tag in cases:
        for name, order in _orders(rays).items():
            for only in (False, True):
                p64 = order.long()
                kw = dict(face_grad_mask=getattr(args0, "face_grad_mask", None),
                          cluster_order=ops.cluster_order(faces), coherent_rays=True)
                if getattr(args0, "n_table", None) is not None:
                    args = ops.Scene3DArgs(faces, args0.catagory, mat_in=args0.mat_in, mat_out=args0.mat_out,
                                           n_table=args0.n_table[:, p64].contiguous(), **kw)
                else:
                    args = ops.Scene3DArgs(faces, args0.catagory, n_in=args0.n_in_arg, n_out=args0.n_out_arg, **kw)
                args.eps = eps
                args.coherent_only = only
                raw = ops.trace3d(rays[:, p64].contiguous(), faces, args, max_passes=4, flags=flags,
                                  new_ray_length=L)
                out = ops.restore_order(raw, order)
                assert np.array_equal(out["counts"], want["counts"]), (tag, name, only)
                for cls in ("finished", "active", "dead", "stopped", "unfinished"):
                    assert torch.equal(out[cls + "_id"], want[cls + "_id"]), (tag, name, only, cls)
                    assert torch.equal(out[cls], want[cls]), (tag, name, only, cls)


def test_engine_coherent_order_is_invisible():
    """OpticalEngine(coherent=True / "auto"): ray_trace() runs over the sorted source (ordered on
    the device, restored inside the trace's autograd node) and hands back the same ray sets as the
    natural-order trace, inherited fields included."""
    import bench
    outs = {}
    for mode in (False, True, "auto"):
        eng, system, params = bench.build_scene(60_000, 20, 6, torch.float32)
        eng.coherent = mode
        eng.ray_trace(3)
        eng.ray_trace(3)
        assert (getattr(eng, "_order_cache", None) is not None) == (mode is not False)
        fin = eng.finished_rays
        outs[mode] = (torch.stack([fin[f] for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")]),
                      fin["wavelength"], fin["object_coords"], eng.last_trace["finished_id"])
    for k in range(4):
        assert torch.equal(outs[True][k], outs[False][k]) and torch.equal(outs["auto"][k], outs[False][k])
