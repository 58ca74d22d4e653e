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
