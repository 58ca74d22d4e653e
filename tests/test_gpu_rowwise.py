"""
RowwiseError: an arbitrary ROW-WISE torch error function of the finished rays on the optimiser's
fixed-shape path (tfrt_scene3d.in_place == 2: every source ray's column, a mask for the rays that
finished; fn's own torch kernels and their autograd inside the step's HIP graph) against the same
function on the generic path (ray sets cut with the counts read back, tfrt/optimizer.py:216-220's
tape as torch autograd) and against the built-in GoalError kernel.
"""
import numpy as np
import pytest
import torch

from test_gpu_engine import _build_lens

pytestmark = pytest.mark.gpu


def _erf(rays):
    # two error terms per ray with different weights, geometry and an inherited source field,
    # a non-polynomial function (element-wise: row i from row i)
    dy = rays["y_end"].double() + rays["object_coords"][:, 1]
    dz = rays["z_end"].double() + rays["object_coords"][:, 2]
    return torch.stack([dy ** 2, 2.0 * torch.log1p(dz ** 2)], dim=1)


def _make(n_rays, mode, fn=_erf, ray_dtype=torch.float64, in_place="auto"):
    import tfrt.optimizer as optimizer
    eng, system, lens, target, source = _build_lens(n_rays, k=3, ray_dtype=ray_dtype)
    eng.in_place = in_place
    erf = optimizer.RowwiseError(fn) if mode != "plain" else (lambda engine: fn(engine.finished_rays))
    opt = optimizer.SGD_Optimizer(eng, lens.parameters, erf, 3, learning_rate=3e-4, grad_clip=1e9,
                                  fused=False if mode in ("generic", "plain") else "auto",
                                  graph="auto" if mode == "graph" else False, speculative=False)
    opt.suppress_warnings = True
    return opt, eng, lens


def _run(opt, steps, lrs):
    return [float(opt.single_step(None, lr_scale=lrs[i])) for i in range(steps)]


def test_rowwise_error_on_the_fixed_shape_path_equals_the_generic_path():
    steps = 10
    lrs = list(np.linspace(1.0, 0.4, steps))
    runs = {}
    for mode in ("plain", "generic", "eager", "graph"):
        opt, eng, lens = _make(6000, mode)
        errs = _run(opt, steps, lrs)
        runs[mode] = (errs, [p.detach().cpu().clone() for p in lens.parameters], opt, eng)
    ref_err, ref_p = runs["plain"][0], runs["plain"][1]
    assert ref_err[-1] < ref_err[0]
    assert runs["generic"][2]._fused_step is None
    for mode in ("eager", "graph"):
        fs = runs[mode][2]._fused_step
        assert fs is not None and fs.in_place and fs.steps >= steps - 3, (mode, fs)
        assert fs.capture_error is None, fs.capture_error
    assert runs["graph"][2]._fused_step.graph_replays >= 2
    for mode in ("generic", "eager", "graph"):
        np.testing.assert_allclose(runs[mode][0], ref_err, rtol=1e-10, atol=0, err_msg=mode)
        for a, b in zip(runs[mode][1], ref_p):
            assert float((a - b).abs().max()) <= 1e-11, mode
    # the ray sets of the last step, cut lazily from the in-place tape, are the generic path's
    fin_g = runs["generic"][3].finished_rays
    fin_f = runs["graph"][3].finished_rays
    for f in ("x_start", "y_end", "z_end", "object_coords"):
        np.testing.assert_allclose(fin_f[f].detach().cpu().numpy(), fin_g[f].detach().cpu().numpy(),
                                   rtol=0, atol=1e-9, err_msg=f)
    assert int(float(runs["graph"][2].last_error_terms)) == 2 * fin_g["y_end"].shape[0]


def test_rowwise_error_equals_the_goal_error_kernel_for_the_same_error():
    """(out - goal)^2 stated as a RowwiseError and as a GoalError: same errors, same updates."""
    import tfrt.optimizer as optimizer

    def sq(rays):
        return torch.stack([(rays["y_end"].double() + rays["object_coords"][:, 1]) ** 2,
                            (rays["z_end"].double() + rays["object_coords"][:, 2]) ** 2], dim=1)

    steps = 8
    lrs = [1.0] * steps
    opt_r, eng_r, lens_r = _make(5000, "graph", fn=sq, ray_dtype=torch.float32)
    errs_r = _run(opt_r, steps, lrs)
    eng, system, lens, target, source = _build_lens(5000, k=3, ray_dtype=torch.float32)
    goal = optimizer.GoalError(("y_end", "z_end"), lambda src: -src["object_coords"][:, 1:])
    opt_g = optimizer.SGD_Optimizer(eng, lens.parameters, goal, 3, learning_rate=3e-4, grad_clip=1e9,
                                    speculative=False)
    opt_g.suppress_warnings = True
    errs_g = _run(opt_g, steps, lrs)
    assert opt_r._fused_step.graph_replays >= 1 and opt_r._fused_step.in_place
    np.testing.assert_allclose(errs_r, errs_g, rtol=2e-6)
    for a, b in zip(lens_r.parameters, lens.parameters):
        assert float((a - b).detach().abs().max()) <= 1e-7 * max(1.0, float(b.detach().abs().max()))


def test_rowwise_error_without_an_in_place_trace_takes_the_generic_path():
    opt, eng, lens = _make(5000, "graph", in_place=False)
    errs = _run(opt, 5, [1.0] * 5)
    assert opt._fused_step is None and all(np.isfinite(errs))
