"""
The fused optimiser step (fused_step.FusedStep: fixed launch sequence, built-in GoalError kernel,
optional HIP-graph replay) against the generic path (arbitrary error function through torch
autograd) and against the oracle.  Same scene, same GoalError: errors, parameters and ray sets
must agree; float64 atomics make the last bits of a gradient sum order-dependent, hence 1e-12
relative instead of bit equality.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tracer
from test_gpu_engine import _build_lens, _oracle_for

pytestmark = pytest.mark.gpu


def _goal(src):
    return -src["object_coords"][:, 1:]


def _make(n_rays, mode, k=3, ray_dtype=torch.float32, accumulators=False, **kw):
    """mode: 'generic' (fused=False), 'eager' (fused, no graph), 'graph'."""
    import tfrt.optimizer as optimizer
    eng, system, lens, target, source = _build_lens(n_rays, k=k, ray_dtype=ray_dtype)
    erf = optimizer.GoalError(("y_end", "z_end"), _goal)
    opt = optimizer.SGD_Optimizer(
        eng, lens.parameters, erf, 3, learning_rate=3e-4, grad_clip=1e9,
        fused=False if mode == "generic" else "auto", graph="auto" if mode == "graph" else False,
        speculative=False, **kw)
    opt.suppress_warnings = True
    acc = None
    if accumulators:
        import tfrt.mesh_tools as mt
        _, a = mt.mesh_parametrization_tools(lens.surfaces[0].zero_points, 0)
        acc = [torch.as_tensor(np.asarray(a)), None]
    return opt, eng, system, lens, target, source, acc


def _run(opt, acc, steps, lrs=None):
    errs = []
    for i in range(steps):
        errs.append(float(opt.single_step(acc, lr_scale=1.0 if lrs is None else lrs[i])))
    return errs


def _params(lens):
    return [p.detach().cpu().clone() for p in lens.parameters]


@pytest.mark.parametrize("accumulators", [False, True])
def test_fused_and_graph_steps_equal_the_generic_path(accumulators):
    steps = 8
    lrs = list(np.linspace(1.0, 0.3, steps))           # a learning-rate schedule (optimizer.py:384)
    runs = {}
    for mode in ("generic", "eager", "graph"):
        # (float64 ray state: with float32 rows the generic path rounds the gradient seeds to
        # float32 on their way through autograd's dtype casts, the fused kernel does not)
        opt, eng, system, lens, *_rest, acc = _make(2000, mode, accumulators=accumulators,
                                                    ray_dtype=torch.float64)
        runs[mode] = (_run(opt, acc, steps, lrs), _params(lens), opt)
    ref_err, ref_p, _ = runs["generic"]
    assert ref_err[-1] < ref_err[0]                     # the lens is being optimised
    assert runs["generic"][2]._fused_step is None
    assert runs["eager"][2]._fused_step.graph_replays == 0
    g = runs["graph"][2]._fused_step
    assert g.capture_error is None, g.capture_error
    assert g.graph_replays >= steps - 4                 # warm-up steps run eagerly, then replays
    for mode in ("eager", "graph"):
        err, p, _ = runs[mode]
        np.testing.assert_allclose(err, ref_err, rtol=1e-11, atol=0, err_msg=mode)
        for a, b in zip(p, ref_p):
            assert float((a - b).abs().max()) <= 1e-12, mode


def test_fused_step_gradient_against_oracle_autograd():
    """One fused step with an inactive clip: (p_before - p_after) / (0.01 * scale) is the summed
    gradient, compared with torch.autograd through the oracle on the parameters the trace used
    (north-star tolerance for float32 ray state: 1e-5 relative)."""
    opt, eng, system, lens, target, source, _ = _make(1500, "eager")
    system.update()                                      # constraints settle the parameters
    used = _params(lens)
    err = float(opt.single_step(None))
    after = _params(lens)
    scale = 0.01 * opt.learning_rate
    q = [u.clone().requires_grad_(True) for u in used]
    osys, src = _oracle_for(system, lens, target, source, q)
    for k in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"):
        src[k] = src[k].float().double()
    ref = tracer.ray_trace(osys, src, max_iterations=3, inherit=("wavelength", "object_coords"))
    rf = ref["finished"]
    rerr = (torch.stack([rf["y_end"], rf["z_end"]], 1) + rf["object_coords"][:, 1:]) ** 2
    rg = torch.autograd.grad(rerr.sum(), q)
    assert abs(err - float(rerr.mean().detach())) <= 1e-5 * float(rerr.mean().detach())
    assert int(float(opt.last_error_terms)) == rerr.numel()
    # the second surface's constraint (thickness vs the first surface, boundaries.py:208-215)
    # shifts it after the first one moved; compare the first surface, which only the SGD step moves
    g0 = (used[0] - after[0]) / scale
    # (ThicknessConstraint(0, "min") on surface 0 re-runs at the NEXT update, not after apply)
    rel = float((g0 - rg[0]).abs().max() / rg[0].abs().max())
    assert rel < 1e-5, f"gradient rel err {rel:.2e}"
    g1 = (used[1] - after[1]) / scale
    rel = float((g1 - rg[1]).abs().max() / rg[1].abs().max())
    assert rel < 1e-5, f"gradient rel err {rel:.2e}"


def test_ray_sets_after_a_fused_step_are_cut_lazily_and_equal_a_plain_trace():
    opt, eng, system, lens, *_ = _make(1200, "graph", ray_dtype=torch.float64)
    for _ in range(6):
        opt.single_step(None)
    assert opt._fused_step.graph_replays >= 2
    assert eng._pending_trace is not None               # nothing was read back during the steps
    before = _params(lens)
    # the sets describe the trace of the last step, i.e. the parameters BEFORE its update; redo
    # that step's trace by hand on a second, identical engine state
    fin = {f: eng.finished_rays[f].detach().clone() for f in ("x_start", "y_end", "z_end", "wavelength",
                                                              "object_coords")}
    assert eng._pending_trace is None
    counts = eng.last_trace["counts"]
    assert int(counts[:, 1].sum()) == fin["y_end"].shape[0] > 900
    # a generic trace with the parameters rolled back one step reproduces them: run the same six
    # steps on the generic path and stop before the last apply
    opt2, eng2, system2, lens2, *_ = _make(1200, "generic", ray_dtype=torch.float64)
    for _ in range(5):
        opt2.single_step(None)
    system2.update()
    eng2.ray_trace(3)
    for f, v in fin.items():
        np.testing.assert_allclose(v.cpu().numpy(), eng2.finished_rays[f].detach().cpu().numpy(),
                                   rtol=0, atol=1e-10, err_msg=f)
    assert all(torch.isfinite(p).all() for p in before)


def test_goal_error_kernel_sum_is_reproducible_and_matches_torch():
    """tfrt_goal_error3d through ctypes: fixed-order error sum (bit-identical from run to run),
    seed rows = 2 (output - goal), untouched rows stay zero."""
    from tensorflowraytrace_amd import _lib, ops
    L = _lib.lib()
    dev = "cuda:0"
    cap, n, n_src = 70_001, 61_234, 90_000
    g = torch.Generator(device=dev).manual_seed(3)
    fin = torch.randn((6, cap), dtype=torch.float32, device=dev, generator=g)
    ids = torch.randint(0, n_src, (cap,), dtype=torch.int32, device=dev, generator=g)
    goal = torch.randn((2, n_src), dtype=torch.float64, device=dev, generator=g)
    P = 3
    counts = torch.zeros(8 * (P + 1), dtype=torch.int32, device=dev)
    counts[8 * P + 1] = n                               # total_finished
    counts[8 * P + 4], counts[8 * P + 5] = 123456, 2    # n_tests = 2 * 2^32 + 123456
    fields = (ctypes.c_int32 * 6)(4, 5, 0, 0, 0, 0)
    wsb = L.tfrt_goal_error3d_workspace_bytes(cap)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    tests_total = torch.zeros(1, dtype=torch.int64, device=dev)
    outs = []
    goal_by_ray = goal.t().contiguous()      # the other layout: one row per source ray
    for k in range(3):
        g_fin = torch.zeros((6, cap), dtype=torch.float64, device=dev)
        err = torch.zeros(3, dtype=torch.float64, device=dev)
        junk = torch.ones(12345, dtype=torch.float64, device=dev)
        table, col_stride, ray_stride = (goal, n_src, 1) if k < 2 else (goal_by_ray, 1, 2)
        _lib.check(L.tfrt_goal_error3d(ops._p(fin), cap, ops._p(ids), _lib.F32, ops._p(counts), P,
                                       fields, 2, ops._p(table), col_stride, ray_stride,
                                       ops._p(g_fin), ops._p(err),
                                       ops._p(junk), 12000, ops._p(tests_total), ops._p(ws), wsb,
                                       ops._stream(fin)), "tfrt_goal_error3d")
        outs.append((g_fin, err.cpu()))
        assert not bool(junk[:12000].any()) and bool((junk[12000:] == 1).all())
    assert int(tests_total.item()) == 3 * (2 * 2 ** 32 + 123456)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[1][1], outs[2][1])
    assert torch.equal(outs[0][0], outs[2][0])           # (either layout of the goal table)
    r = fin[4:6, :n].double() - goal[:, ids[:n].long()]
    want = (r ** 2).sum()
    g_fin, err = outs[0]
    assert abs(float(err[0]) - float(want)) <= 1e-12 * float(want)
    assert float(err[1]) == 2 * n and abs(float(err[2]) - float(want) / (2 * n)) <= 1e-15
    assert torch.equal(g_fin[4:6, :n], 2.0 * r)
    assert not bool(g_fin[:4].any()) and not bool(g_fin[4:6, n:].any())


def test_folded_reverse_sweep_equals_goal_error_plus_backward():
    """tfrt_trace3d_backward_goal (error, seed and the whole reverse sweep in one launch) against
    tfrt_goal_error3d + tfrt_trace3d_backward on the same coherent trace: the same error to the
    last bits of a differently ordered sum, the same parameters after eight steps; float32 ray
    state too (the residuals are formed from the finished rows AS STORED in both)."""
    import tensorflowraytrace_amd.fused_step as fs
    # (the face sums are float64 atomics: their last bits differ from run to run, and eight steps
    # of this lively optimisation amplify that to ~1e-12)
    for ray_dtype, tol in ((torch.float64, 1e-10), (torch.float32, 1e-8)):
        runs = {}
        for fold in (True, False):
            opt, eng, system, lens, *_rest, acc = _make(20000, "graph", k=6, ray_dtype=ray_dtype)
            eng.coherent = True
            old = fs.FusedStep.fold_backward
            fs.FusedStep.fold_backward = fold
            try:
                errs = _run(opt, None, 8)
            finally:
                fs.FusedStep.fold_backward = old
            g = opt._fused_step
            assert g.capture_error is None and g.graph_replays >= 2 and g.folded_backward == fold
            runs[fold] = (errs, _params(lens))
        np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=tol, atol=0)
        for a, b in zip(runs[True][1], runs[False][1]):
            assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_fused_step_in_coherent_order_equals_the_generic_path():
    """From 4096 rays on the engine sorts the source along a Hilbert curve (``coherent='auto'``:
    tfrt_ray_order, k_intersect_beam, per-wavefront face sums).  The
    sorted trace must be invisible: same errors and parameters as the generic natural-order path,
    and the lazily cut ray sets come back in the reference's order."""
    steps = 8
    runs = {}
    for mode in ("generic", "graph"):
        opt, eng, system, lens, *_rest, acc = _make(20000, mode, k=6, ray_dtype=torch.float64)
        if mode == "generic":
            eng.coherent = False              # (the reference's order all the way: the yardstick)
        else:
            # ("auto" would go back to natural order after three steps: a tenth of this coarse
            # lens's wavefronts are no narrow bundles)
            eng.coherent = True
        errs = _run(opt, None, steps)
        runs[mode] = (errs, _params(lens), opt, eng)
    g = runs["graph"][2]._fused_step
    assert g.capture_error is None and g.graph_replays >= 2
    assert getattr(runs["graph"][3], "_order_cache", None) is not None      # the sorted source
    assert g.folded_backward        # error + seed + reverse sweep in one launch (tfrt_trace3d_backward_goal)
    assert getattr(runs["generic"][3], "_order_cache", None) is None
    np.testing.assert_allclose(runs["graph"][0], runs["generic"][0], rtol=1e-10, atol=0)
    for a, b in zip(runs["graph"][1], runs["generic"][1]):
        assert float((a - b).abs().max()) <= 1e-11
    # ray sets of the last step: fused + sorted (restored lazily) against a natural-order trace of
    # the same parameters
    eng = runs["graph"][3]
    fin = {f: eng.finished_rays[f].detach().clone() for f in ("x_start", "y_end", "z_end", "wavelength",
                                                              "object_coords")}
    ids = eng.last_trace["finished_id"].clone()
    opt2, eng2, system2, lens2, *_ = _make(20000, "generic", k=6, ray_dtype=torch.float64)
    for _ in range(steps - 1):
        opt2.single_step(None)
    system2.update()
    eng2.ray_trace(3)
    assert torch.equal(ids, eng2.last_trace["finished_id"])
    # (the two runs' parameters differ by ~1e-11 after seven steps; a ray's end point on the target
    # ten units away moves by a few hundred times that)
    for f, v in fin.items():
        np.testing.assert_allclose(v.cpu().numpy(), eng2.finished_rays[f].detach().cpu().numpy(),
                                   rtol=0, atol=1e-8, err_msg=f)
