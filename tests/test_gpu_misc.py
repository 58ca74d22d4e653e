"""Device-side odds and ends next to the hot path: the seeded per-device generator of the Random*
distributions and the analysis helpers on device tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_random_distributions_draw_on_the_device_and_repeat_with_the_seed():
    import tensorflowraytrace_amd as tfa
    import tensorflowraytrace_amd.distributions as distributions
    tfa.set_device("cuda:0")
    distributions.seed(11)
    a = distributions.RandomUniformCircle(5000, 0.7)
    a.update()
    p1 = a.points.clone()
    a.update()
    p2 = a.points.clone()
    assert p1.is_cuda and p1.shape == (5000, 2)
    assert not torch.equal(p1, p2)                             # a new draw every update
    assert float(torch.linalg.norm(p1, dim=1).max()) <= 0.7 + 1e-12
    distributions.seed(11)
    b = distributions.RandomUniformCircle(5000, 0.7)
    b.update()
    assert torch.equal(b.points, p1)                           # same seed, same stream
    distributions.seed(12)
    b.update()
    assert not torch.equal(b.points, p1)
    # uniform over the disc: mean radius 2/3 R, quadrant counts balanced
    r = torch.linalg.norm(p1, dim=1)
    assert abs(float(r.mean()) - 2 / 3 * 0.7) < 0.01
    q = ((p1[:, 0] > 0).long() * 2 + (p1[:, 1] > 0).long()).bincount(minlength=4).cpu().numpy()
    assert np.all(np.abs(q - 1250) < 150)


def test_histogram2d_and_distribution_differential_on_device_tensors():
    import tensorflowraytrace_amd as tfa
    import tensorflowraytrace_amd.analyze as analyze
    tfa.set_device("cuda:0")
    rng = np.random.default_rng(3)
    x, y = rng.normal(0, 0.4, 200_000), rng.normal(0, 0.4, 200_000)
    xd, yd = torch.tensor(x).cuda(), torch.tensor(y).cuda()
    H = analyze.histogram2D(xd, yd, ((-1, 1), (-1, 1)), x_bins=32, y_bins=24)
    assert H.is_cuda and H.shape == (24, 32)
    inside = (np.abs(x) < 1) & (np.abs(y) < 1)
    want, _, _ = np.histogram2d(y[inside], x[inside], bins=(24, 32), range=((-1, 1), (-1, 1)))
    got = H.cpu().numpy().astype(np.int64)
    # interior bins agree exactly; the edge bins also collect the out-of-range points
    np.testing.assert_array_equal(got[1:-1, 1:-1], want[1:-1, 1:-1].astype(np.int64))
    assert got.sum() == 200_000
    goal = lambda gx, gy: torch.exp(-(gx ** 2 + gy ** 2) / (2 * 0.4 ** 2))     # noqa: E731
    dd = analyze.DistributionDifferential(goal, ((-1, 1), (-1, 1)), x_bins=24)
    good = float(dd(xd, yd))
    flat = float(dd(torch.tensor(rng.uniform(-1, 1, 200_000)).cuda(),
                    torch.tensor(rng.uniform(-1, 1, 200_000)).cuda()))
    assert dd.saved_histo.is_cuda and 0 <= good < 0.01 < flat
