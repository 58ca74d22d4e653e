"""Host logic of the Lambertian / arbitrary-density / stored base-point distributions
(tfrt/distributions.py:394-556, 1778-2011, 2123-3195).  The reference has no tests for them:
these check the defining properties stated in its docstrings."""
import math
import pickle

import numpy as np
import pytest
import torch

import tfrt.distributions as dist


def test_lambertian_angles_have_evenly_spaced_sines():
    d = dist.StaticLambertianAngularDistribution(-0.7, 1.1, 9)
    ranks = d.ranks.cpu().numpy()
    np.testing.assert_allclose(np.diff(ranks), np.diff(ranks)[0], rtol=1e-12)
    np.testing.assert_allclose(ranks[[0, -1]], [math.sin(-0.7), math.sin(1.1)], rtol=1e-12)
    np.testing.assert_allclose(np.sin(d.angles.cpu().numpy()), ranks, rtol=1e-12, atol=1e-15)
    with pytest.raises(ValueError):
        dist.StaticLambertianAngularDistribution(-2.0, 1.0, 5)     # outside [-pi/2, pi/2]


def test_random_lambertian_angles_stay_in_range_and_resample():
    dist.seed(5)
    d = dist.RandomLambertianAngularDistribution(-0.5, 0.25, 2000)
    a = d.angles.cpu().numpy()
    assert a.min() >= -0.5 and a.max() <= 0.25
    # uniform in sin(angle): mean of the ranks is the midpoint of the sine limits
    assert abs(d.ranks.mean().item() - 0.5 * (math.sin(-0.5) + math.sin(0.25))) < 0.02
    first = a.copy()
    d.update()
    assert not np.array_equal(first, d.angles.cpu().numpy())


def test_lambertian_sphere_spacing_and_ranks():
    s = dist.StaticLambertianSphere(1.2, 50, radius=2.0)
    pts = s.points.cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(pts, axis=1), 2.0, rtol=1e-12)
    cos2 = (pts[:, 0] / 2.0) ** 2
    np.testing.assert_allclose(np.diff(cos2), np.diff(cos2)[0], rtol=1e-9)   # even in cos^2(phi)
    np.testing.assert_allclose(cos2[[0, -1]], [1.0, math.cos(1.2) ** 2], atol=1e-12)
    ranks = s.ranks.cpu().numpy()                                           # (phi, theta mod 2pi)
    np.testing.assert_allclose(np.cos(ranks[:, 0]), pts[:, 0] / 2.0, atol=1e-12)
    assert ranks[:, 1].min() >= 0 and ranks[:, 1].max() < 2 * math.pi
    u = dist.StaticUniformSphere(1.2, 50)
    c = u.points.cpu().numpy()[:, 0]
    np.testing.assert_allclose(np.diff(c), np.diff(c)[0], rtol=1e-9)        # even in cos(phi)
    dist.seed(1)
    r = dist.RandomLambertianSphere(0.9, 500)
    assert torch.acos(r.points[:, 0]).max().item() <= 0.9 + 1e-12


def test_arbitrary_distribution_follows_the_density():
    density = lambda x, y: 1.0 + 3.0 * (x > 0)          # right half four times as dense
    ad = dist.ArbitraryDistribution(density, ((-1.0, 1.0, 64), (-1.0, 1.0, 64)))
    rng = np.random.default_rng(0)
    ux, uy = rng.uniform(-1, 1, 40000), rng.uniform(-1, 1, 40000)
    x, y = ad(ux, uy)
    assert x.min() >= -1 and x.max() <= 1 and y.min() >= -1 and y.max() <= 1
    frac_right = (x > 0).mean()
    assert abs(frac_right - 0.8) < 0.02
    assert abs(np.mean(y)) < 0.02                        # y density is flat
    # monotone in x: the map preserves order along x
    order = np.argsort(ux)
    assert np.all(np.diff(x[order]) >= -1e-12)
    with pytest.raises(ValueError):
        dist.ArbitraryDistribution(lambda x, y: -np.ones_like(x), ((-1, 1, 8), (-1, 1, 8)))
    with pytest.raises(ValueError):
        dist.ArbitraryDistribution(np.zeros((4, 4)), ((-1, 1), (-1, 1)))   # zero slice
    with pytest.raises(ValueError):
        dist.ArbitraryDistribution(np.ones(4), ((-1, 1), (-1, 1)))         # not 2-D


def test_flatten_distribution_undoes_a_dense_blob():
    rng = np.random.default_rng(1)
    x = np.clip(rng.normal(0, 0.3, 50000), -1, 1)
    y = np.clip(rng.normal(0, 0.3, 50000), -1, 1)
    fx, fy = dist.flatten_distribution(x, y, ((-1, 1, 40), (-1, 1, 40)))
    assert fx.min() >= 0 and fx.max() <= 1
    hist, _ = np.histogram(fx, bins=10, range=(0, 1))
    assert hist.max() / hist.min() < 1.3                 # x marginal is flat afterwards
    raw, _ = np.histogram(x, bins=10, range=(-1, 1))
    assert raw.max() / max(raw.min(), 1) > 50            # and was not before


def test_cumulative_density_function_forward_and_inverse_are_inverse_maps():
    rng = np.random.default_rng(2)
    dens = rng.random((12, 12)) + 0.2
    cdf = dist.CumulativeDensityFunction(((-1.0, 1.0), (-2.0, 2.0)), density=dens)
    assert (cdf.x_res, cdf.y_res) == (12, 12)
    u = rng.random((500, 2)) * 0.98 + 0.01
    mapped = cdf(u)
    assert mapped[:, 0].min() >= -1 and mapped[:, 0].max() <= 1
    assert mapped[:, 1].min() >= -2 and mapped[:, 1].max() <= 2
    # y is mapped independently of x: icdf_y(cdf_y(u)) = u
    np.testing.assert_allclose(cdf._y_icdf(cdf._y_cdf(u[:, 1])), u[:, 1], atol=1e-6)
    with pytest.raises(RuntimeError):
        dist.CumulativeDensityFunction(((-1, 1), (-1, 1))).compute()
    only_inverse = dist.CumulativeDensityFunction(((-1, 1), (-1, 1)), density=dens,
                                                  direction="inverse")
    with pytest.raises(RuntimeError):
        only_inverse.cdf(u)
    with pytest.raises(ValueError):
        cdf.compute(direction="sideways")
    cdf.accumulate_density(dens)                          # accumulates in place
    np.testing.assert_allclose(cdf._density, 2 * dens.astype(np.float32), rtol=1e-6)


def test_arbitrary_base_points_and_etendue():
    dist.seed(11)
    ad = dist.ArbitraryDistribution(lambda x, y: np.exp(-4 * (x * x + y * y)) + 1e-6,
                                    ((-1, 1, 48), (-1, 1, 48)))
    flat = dist.ArbitraryDistribution(lambda x, y: np.ones_like(x), ((-1, 1, 48), (-1, 1, 48)))
    bp = dist.ArbitraryBasePoints(ad, 3000, rank_distribution=flat)
    assert bp.points.shape == (3000, 2) and bp.ranks.shape == (3000, 2)
    mean_p = torch.linalg.norm(bp.points, dim=1).mean().item()
    mean_r = torch.linalg.norm(bp.ranks, dim=1).mean().item()
    assert abs(mean_p - mean_r) < 1e-9 * max(mean_p, 1)  # etendue: equal mean radius
    assert bp.rank_scale_factor < 1                      # blob is tighter than the flat ranks
    fixed = dist.ArbitraryBasePoints(ad, 100, rank_distribution=flat, auto_reroll=False)
    before = fixed.points.clone()
    fixed.update()
    assert torch.equal(before, fixed.points)             # no reroll: same seeds, same points
    with pytest.raises(ValueError):
        dist.ArbitraryBasePoints(ad, 0)


def test_transform_map_is_a_permutation_and_optimal_beats_greedy():
    rng = np.random.default_rng(4)
    fixed, mutable = rng.random((40, 2)), rng.random((40, 2))
    best = dist.transform_map(fixed, mutable)
    greedy = dist.transform_map_old(fixed, mutable)
    for out in (best, greedy):
        assert sorted(map(tuple, out)) == sorted(map(tuple, mutable))
    cost = lambda m: np.linalg.norm(fixed - m, axis=1).sum()
    assert cost(best) <= cost(greedy) + 1e-12
    assert cost(best) < cost(mutable)
    with pytest.raises(ValueError):
        dist.transform_map(fixed, mutable[:-1])
    with pytest.raises(ValueError):
        dist.transform_map_old(fixed, mutable, origin=np.zeros(3))


def test_image_base_points_put_grey_level_many_points_in_each_pixel(tmp_path):
    img = np.array([[0, 50, 50], [200, 0, 200]], dtype=float)   # grey levels -> 0, 1, 2 points
    dist.seed(8)
    ibp = dist.ImageBasePoints(img, 3.0, 2.0)
    assert (ibp.x_res, ibp.y_res, ibp.grey_levels) == (2, 3, 3)
    pts = ibp.points.cpu().numpy()
    assert pts.shape == (0 + 1 + 1 + 2 + 0 + 2, 2)
    xe, ye = np.linspace(-1.5, 1.5, 3), np.linspace(-1.0, 1.0, 4)
    hist, _, _ = np.histogram2d(pts[:, 0], pts[:, 1], bins=(xe, ye))
    np.testing.assert_array_equal(hist, [[0, 1, 1], [2, 0, 2]])
    # the same through an image file
    from PIL import Image
    path = str(tmp_path / "levels.png")
    Image.fromarray(img.astype(np.uint8), mode="L").save(path)
    from_file = dist.ImageBasePoints(path, 3.0, 2.0)
    assert from_file.points.shape == ibp.points.shape
    with pytest.raises(ValueError):
        dist.ImageBasePoints(img, -1.0)


def test_precompiled_base_points_pickle_format_and_resampling(tmp_path):
    dist.seed(3)
    base = dist.StaticUniformSquare(2.0, 5)
    pre = dist.PrecompiledBasePoints(base, sample_count=7)
    assert pre.sampling_domain_size == 25 and pre.points.shape == (7, 2)
    assert pre.ranks.shape == (7, 2)
    full = base.points.cpu().numpy()
    for p in pre.points.cpu().numpy():                   # sampled with replacement from the base
        assert np.isclose(full, p).all(axis=1).any()
    path = str(tmp_path / "points.pkl")
    pre.save(path)
    with open(path, "rb") as f:
        blob = pickle.load(f)                            # the reference's on-disk layout
    assert set(blob) == {"points", "ranks"} and isinstance(blob["points"], np.ndarray)
    np.testing.assert_array_equal(blob["points"], full)
    again = dist.PrecompiledBasePoints(path, do_downsample=False, perturbation=[0.0, 0.5])
    moved = again.points.cpu().numpy() - full
    assert np.all(moved[:, 0] == 0) and moved[:, 1].std() > 0.1
    with pytest.raises(ValueError):
        dist.PrecompiledBasePoints(path, perturbation=[1.0, 2.0, 3.0])
    again.clear()
    assert again.points is None and again.sampling_domain_size == 0


def test_square_rank_lambertian_sphere():
    dist.seed(21)
    s = dist.SquareRankLambertianSphere(4000, angular_cutoff=0.6, sampling_resolution=96)
    pts = s.points.cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(pts, axis=1), 1.0, rtol=1e-12)
    phi = np.arccos(np.clip(pts[:, 0], -1, 1))
    assert phi.max() <= 0.6 + 0.03                       # grid resolution of the disc edge
    ranks = s.ranks.cpu().numpy()
    assert ranks.min() >= -1 and ranks.max() <= 1 and abs(ranks.mean()) < 0.05
    # Lambertian: sin^2(phi) is uniform on [0, sin^2(cutoff)]
    q = np.sin(phi) ** 2 / math.sin(0.6) ** 2
    hist, _ = np.histogram(q, bins=5, range=(0, 1))
    assert hist.max() / hist.min() < 1.25
    with pytest.raises(ValueError):
        dist.SquareRankLambertianSphere(10, angular_cutoff=2.0)
