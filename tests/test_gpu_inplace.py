"""
The in-place trace (tfrt_scene3d.in_place: all passes in ONE launch, every ray kept in its slot,
ray sets compacted afterwards by a scan + a gather from the tape) against the per-pass launch
sequence (intersect -> react per pass, children compacted between passes -- the reference's
boolean_mask order, tfrt/engine.py:2069-2111, which tests/test_gpu_stress.py and
tests/test_reference_golden.py pin against the oracle and the reference fixtures).

Everything discrete and every ray must be IDENTICAL (a ray's nearest hit does not depend on which
rays share its wavefront; children are rounded to the state type in the same place); gradients
agree to the last bits of a differently ordered float64 sum.
"""
import ctypes

import numpy as np
import pytest
import torch

import scene_util
import test_gpu_stress as st
from test_gpu_trace3d import _gpu_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CLASSES = ("finished", "active", "dead", "stopped", "unfinished")


def _flags():
    from tensorflowraytrace_amd import _lib
    return _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED


def _same(out, ref, tag):
    assert np.array_equal(out["counts"], ref["counts"]), tag
    assert out["n_tests"] == ref["n_tests"], tag
    for cls in CLASSES:
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (tag, cls)
        assert torch.equal(out[cls], ref[cls]), (tag, cls)
        if cls != "unfinished":
            assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), (tag, cls)


def _lens_args(sc, fv, order, in_place, eps=None):
    from tensorflowraytrace_amd import ops
    a = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                        n_table=sc.n_table[:, order.long()].contiguous(),
                        cluster_order=ops.cluster_order(fv), coherent_rays=True)
    if eps is not None:
        a.eps = eps
    a.coherent_only = True
    a.in_place = in_place
    return a


@pytest.mark.parametrize("n_rays,dtype,passes", [
    (3000, torch.float32, 4), (9001, torch.float64, 3), (20000, torch.float16, 5),
    (70000, torch.float32, 12),      # more passes than the reverse sweep's LDS columns hold
    (200000, torch.float32, 3),      # 64-ray wavefronts (<= 160k rays: 32)
    (64, torch.float64, 2), (97, torch.float32, 6),
])
def test_in_place_trace_equals_the_per_pass_trace_on_lens_scenes(n_rays, dtype, passes):
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(n_rays, k_front=7, k_back=5)
    src, fv, sc, _ = _gpu_scene(scene, dtype, cluster="group")
    fv = fv.detach()
    order = ops.ray_order(src)
    rays = src[:, order.long()].contiguous()
    outs = {}
    for in_place in (False, True):
        outs[in_place] = ops.trace3d(rays, fv, _lens_args(sc, fv, order, in_place),
                                     max_passes=passes, flags=_flags())
    assert int(outs[False]["counts"][:, 1].sum()) > (0.5 * n_rays if n_rays >= 1000 else 0)   # most rays finish
    _same(outs[True], outs[False], (n_rays, dtype, passes))
    # ... and, restored, the natural-order trace (the reference's order)
    plain = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out, n_table=sc.n_table,
                            cluster_order=ops.cluster_order(fv))
    ref = ops.trace3d(src, fv, plain, max_passes=passes, flags=_flags())
    _same(ops.restore_order(outs[True], order), ref, ("restored", n_rays, dtype, passes))


@pytest.mark.parametrize("seed", [2, 7, 16, 21, 25, 35, 52, 58])
def test_in_place_trace_on_adversarial_soups(seed):
    """Random triangle soups: no coherence (every wavefront is cut down to single rays), all four
    classes, mirrors, total internal reflection, coplanar ties, unusual epsilons."""
    from tensorflowraytrace_amd import ops
    sc0 = st._soup(seed)
    if sc0["rays"].shape[1] < 64:
        pytest.skip("fewer rays than a wavefront")
    fv = sc0["P"].to(DEV)
    eps = [(1e-10, 1e-10, 1e-10), (1e-10, 1e-3, 1e-7), (1e-10, 0.2, -0.01)][seed % 3]
    base = dict(n_in=sc0["n_in"].to(DEV), n_out=sc0["n_out"].to(DEV))
    for dtype in (torch.float64, torch.float32):
        r = sc0["rays"].to(DEV).to(dtype)
        plain = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), **base)
        plain.eps = eps
        ref = ops.trace3d(r, fv, plain, max_passes=4, flags=_flags(), new_ray_length=sc0["L"],
                          dead_ray_length=0.5 if seed % 2 else None)
        order = ops.ray_order(r)
        args = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv),
                               coherent_rays=True, **base)
        args.eps = eps
        args.coherent_only = args.in_place = True
        raw = ops.trace3d(r[:, order.long()].contiguous(), fv, args, max_passes=4, flags=_flags(),
                          new_ray_length=sc0["L"], dead_ray_length=0.5 if seed % 2 else None)
        _same(ops.restore_order(raw, order), ref, (seed, dtype))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_in_place_gradients_equal_the_per_pass_gradients(dtype):
    """d (an error of the finished, active and stopped rays) / d (face vertices, source rays)
    through the in-place tape and through the per-pass tape, both traced over the same sorted
    rays; class gradients reach the in-place sweep through rec_slot (k_inplace_gather)."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(6000, k_front=6, k_back=4)
    src, fv0, sc, _ = _gpu_scene(scene, dtype, cluster="group")
    order = ops.ray_order(src)
    rays0 = src[:, order.long()].contiguous()
    grads = {}
    for in_place in (False, True):
        fv = fv0.detach().clone().requires_grad_(True)
        rays = rays0.detach().clone().requires_grad_(True)
        out = ops.trace3d(rays, fv, _lens_args(sc, fv.detach(), order, in_place), max_passes=3,
                          flags=_flags())
        w = torch.linspace(0.5, 1.5, out["finished"].shape[1], device=DEV, dtype=torch.float64)
        err = (w * (out["finished"][4].double() ** 2 + out["finished"][5].double() ** 2)).sum()
        err = err + (out["active"][3].double() * out["active"][0].double()).sum() * 1e-3
        if out["dead"].shape[1]:
            err = err + out["dead"][4].double().sum() * 1e-3
        grads[in_place] = torch.autograd.grad(err, [fv, rays])
    for a, b in zip(grads[True], grads[False]):
        scale = float(b.double().abs().max())
        assert scale > 0
        tol = 1e-12 if dtype == torch.float64 else 2e-6
        assert float((a.double() - b.double()).abs().max()) <= tol * scale


def test_compact_entry_fills_the_sets_a_forward_call_without_outputs_left_out():
    """tfrt_trace3d_forward with no room for ray sets (the fused step's call) + tfrt_trace3d_compact
    later == a forward call that was given the outputs."""
    from tensorflowraytrace_amd import ops, _lib
    L = _lib.lib()
    scene = scene_util.lens_scene(5000, k_front=6, k_back=4)
    src, fv, sc, _ = _gpu_scene(scene, torch.float32, cluster="group")
    fv = fv.detach()
    order = ops.ray_order(src)
    rays = src[:, order.long()].contiguous()
    args = _lens_args(sc, fv, order, True)
    P, N, M = 4, rays.shape[1], fv.shape[0]
    ref = ops.trace3d(rays, fv, args, max_passes=P, flags=_flags())
    wsb = L.tfrt_trace3d_workspace_bytes(N, M, P, _lib.F32)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    counts = torch.zeros(_lib.COUNTS_PER_PASS * (P + 1), dtype=torch.int32, device=DEV)
    none = [ops._ray_out(None, None, None) for _ in range(4)]
    stream = ops._stream(rays)
    st_ = args.struct(fv)
    _lib.check(L.tfrt_trace3d_forward(ops._p(rays), N, N, ctypes.byref(st_), 1.0, 0.0, P, _lib.F32,
                                      _flags(), *[ctypes.byref(o) for o in none], None, None,
                                      ops._p(counts), ops._p(ws), wsb, stream), "forward")
    caps = {"finished": N, "active": N * P, "stopped": N, "dead": N}
    bufs = {k: (torch.empty((6, c), dtype=torch.float32, device=DEV),
                torch.empty(c, dtype=torch.int32, device=DEV),
                torch.empty(c, dtype=torch.int32, device=DEV)) for k, c in caps.items()}
    outs = [ops._ray_out(*bufs[k]) for k in ("finished", "active", "stopped", "dead")]
    unf, unf_id = torch.empty((6, N), dtype=torch.float32, device=DEV), torch.empty(N, dtype=torch.int32, device=DEV)
    _lib.check(L.tfrt_trace3d_compact(ops._p(rays), N, N, 0.0, P, _lib.F32, _flags(),
                                      *[ctypes.byref(o) for o in outs], ops._p(unf), ops._p(unf_id),
                                      ops._p(counts), M, None, ops._p(ws), wsb, stream), "compact")
    host = counts.cpu().numpy()
    assert np.array_equal(host[:P * 8].reshape(P, 8), ref["counts"])
    assert host[P * 8 + 6] == 0                                           # no capacity error
    assert (int(np.uint32(host[P * 8 + 4])) | (int(np.uint32(host[P * 8 + 5])) << 32)) == ref["n_tests"]
    for k in ("finished", "active", "stopped", "dead"):
        n = ref[k].shape[1]
        assert torch.equal(bufs[k][0][:, :n], ref[k]), k
        assert torch.equal(bufs[k][1][:n], ref[k + "_id"]), k
        assert torch.equal(bufs[k][2][:n], ref[k + "_face"]), k
    n = ref["unfinished"].shape[1]
    assert torch.equal(unf[:, :n], ref["unfinished"]) and torch.equal(unf_id[:n], ref["unfinished_id"])


def test_fused_step_in_place_equals_the_per_pass_fused_step():
    """The optimiser step over a sorted static source: in-place trace + folded reverse sweep that
    recomputes the finished rows from the tape, against the per-pass trace + the sweep that reads
    the finished block -- error sums bit for bit (same rays, same fixed-order sum), parameters to
    the last bits of the face sums; ray sets cut lazily from the in-place tape equal the per-pass
    ones."""
    from test_gpu_fused_step import _make, _run, _params
    runs = {}
    for in_place in (False, True):
        opt, eng, system, lens, *_ = _make(6000, "graph", ray_dtype=torch.float32)
        eng.in_place = in_place
        errs = _run(opt, None, 9)
        fs = opt._fused_step
        assert fs.capture_error is None, fs.capture_error
        assert fs.graph_replays >= 3
        assert fs.in_place == in_place, (fs.in_place, in_place)
        fin = {f: eng.finished_rays[f].detach().clone() for f in ("x_start", "y_end", "z_end", "object_coords")}
        runs[in_place] = (errs, _params(lens), fin, eng.last_trace["counts"].copy(),
                          int(float(opt.last_error_terms)), int(fs.tests_total))
    a, b = runs[True], runs[False]
    # (the two runs take the same parameters only up to the last bits of the atomically summed
    # gradients: after the first steps the errors agree to rounding, not bit for bit)
    assert a[0][0] == b[0][0]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-9)
    for p, q in zip(a[1], b[1]):
        assert float((p - q).abs().max()) <= 1e-10
    assert np.array_equal(a[3], b[3])
    assert a[4] == b[4] > 0 and a[5] == b[5] > 0     # error terms, ray-face tests of all steps
    for f in a[2]:
        np.testing.assert_allclose(a[2][f].cpu().numpy(), b[2][f].cpu().numpy(), rtol=0, atol=1e-6, err_msg=f)


@pytest.mark.parametrize("n_rays,dtype,passes", [(5000, torch.float32, 4), (70001, torch.float64, 3),
                                                 (200000, torch.float32, 3)])
def test_in_place_trace_hands_the_sets_back_in_the_callers_order(n_rays, dtype, passes):
    """trace3d(permuted rays, perm=order) with scene.in_place: the ray sets come back in the order
    of the UNPERMUTED source (ids in its numbering, every class pass after pass by ascending id --
    engine.py:2069-2111) straight from the compaction (tfrt_scene3d.ray_slot), equal to the
    natural-order trace bit for bit; gradients w.r.t. faces equal the per-pass path's."""
    from tensorflowraytrace_amd import ops
    scene = scene_util.lens_scene(n_rays, k_front=7, k_back=5)
    src, fv0, sc, _ = _gpu_scene(scene, dtype, cluster="group")
    order = ops.ray_order(src)
    rays = src[:, order.long()].contiguous()
    plain = ops.Scene3DArgs(fv0.detach(), sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                            n_table=sc.n_table, cluster_order=ops.cluster_order(fv0.detach()))
    ref = ops.trace3d(src, fv0.detach(), plain, max_passes=passes, flags=_flags())
    grads = {}
    for in_place in (True, False):
        fv = fv0.detach().clone().requires_grad_(True)
        out = ops.trace3d(rays, fv, _lens_args(sc, fv.detach(), order, in_place), max_passes=passes,
                          flags=_flags(), perm=order)
        _same(out, ref, ("own order", in_place, n_rays, dtype))
        w = torch.linspace(0.5, 1.5, out["finished"].shape[1], device=DEV, dtype=torch.float64)
        err = (w * (out["finished"][4].double() ** 2 + out["finished"][5].double() ** 2)).sum()
        err = err + (out["active"][3].double() * out["active"][0].double()).sum() * 1e-3
        grads[in_place], = torch.autograd.grad(err, [fv])
    scale = float(grads[False].abs().max())
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert float((grads[True] - grads[False]).abs().max()) <= tol * scale
