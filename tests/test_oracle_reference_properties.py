"""
Pins the oracle with the reference's OWN test properties for this path, restated because the
originals cannot execute (TF-1 sessions + functions that no longer exist):

* tests/geometry/test_line_intersect_1to1.py:9-98    (common point <= 1e-4, unit square,
                                                       parallel => invalid)
* tests/geometry/test_line_circle_intersect_1to1.py:12-172  (2 / 1 / 0 root families, 1e-6)
* tests/geometry/test_angle_in_interval.py:26-66     (9 x 9 start/stop grid, 11 in + 11 out)
"""
import math

import numpy as np
import pytest
import torch

from oracle import geom

PI = math.pi


def test_intersecting_lines_common_point():
    rng = np.random.default_rng(0)
    count = 100
    common = rng.uniform(-1.0, 1.0, size=[count, 2])
    first = rng.uniform(-10.0, 10.0, size=[count, 2])
    second = rng.uniform(-10.0, 10.0, size=[count, 2])
    p = [rng.uniform(-10.0, 10.0, size=[count, 1]) for _ in range(4)]
    fs = common + p[0] * (first - common)
    fe = common + p[1] * (first - common)
    ss = common + p[2] * (second - common)
    se = common + p[3] * (second - common)
    x, y, valid, u, v = geom.raw_line_intersect(
        fs[:, 0], fs[:, 1], fe[:, 0], fe[:, 1], ss[:, 0], ss[:, 1], se[:, 0], se[:, 1], 1e-10)
    assert bool(valid.all())
    dist = torch.sqrt((x - torch.tensor(common[:, 0])) ** 2 + (y - torch.tensor(common[:, 1])) ** 2)
    assert float(dist.max()) < 1e-4


def test_unit_square_cases():
    first = np.array([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 1.0, 1.0]])
    second = np.array([[0.0, 0.0, 0.0, 1.0], [0.0, 1.0, 1.0, 1.0]])
    want = np.array([[0.0, 0.0], [1.0, 1.0]])
    x, y, valid, u, v = geom.raw_line_intersect(*first.T, *second.T, 1e-10)
    assert bool(valid.all())
    assert np.abs(x.numpy() - want[:, 0]).max() < 1e-4
    assert np.abs(y.numpy() - want[:, 1]).max() < 1e-4


@pytest.mark.parametrize("a,b", [
    ([0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 1.0]),   # degenerate second line
    ([0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 1.0, 1.0]),   # parallel verticals
    ([0.0, 0.0, 1.0, 1.0], [0.0, 1.0, 1.0, 2.0]),   # parallel diagonals
])
def test_parallel_lines_are_invalid(a, b):
    _, _, valid, _, _ = geom.raw_line_intersect(*[np.array([t]) for t in a],
                                                *[np.array([t]) for t in b], 1e-10)
    assert not bool(valid.any())


def _circles(rng, count):
    xc = rng.uniform(-10.0, 10.0, size=count)
    yc = rng.uniform(-10.0, 10.0, size=count)
    r = rng.uniform(0.1, 2.0, size=count)
    return xc, yc, r


def test_circle_two_intersections():
    rng = np.random.default_rng(1)
    count = 100
    xc, yc, r = _circles(rng, count)
    a1, a2 = rng.uniform(0, 2 * PI, size=count), rng.uniform(0, 2 * PI, size=count)
    p1 = np.stack([xc + r * np.cos(a1), yc + r * np.sin(a1)], 1)
    p2 = np.stack([xc + r * np.cos(a2), yc + r * np.sin(a2)], 1)
    sp, ep = rng.uniform(-10, 10, size=[count, 1]), rng.uniform(-10, 10, size=[count, 1])
    start, end = p1 + sp * (p2 - p1), p1 + ep * (p2 - p1)
    plus, minus = geom.raw_line_circle_intersect(start[:, 0], start[:, 1], end[:, 0], end[:, 1],
                                                 xc, yc, r, 1e-10)
    assert bool(plus["valid"].all()) and bool(minus["valid"].all())
    sep = torch.sqrt((plus["x"] - minus["x"]) ** 2 + (plus["y"] - minus["y"]) ** 2)
    assert float(sep.min()) > 1e-6
    small = 0
    for sol in (plus, minus):
        for p in (p1, p2):
            d = torch.sqrt((sol["x"] - torch.tensor(p[:, 0])) ** 2 + (sol["y"] - torch.tensor(p[:, 1])) ** 2)
            small = small + (d < 1e-6).to(torch.int32)
    assert bool((small == 2).all())


def _tangent_like(rng, count, factor):
    xc, yc, r = _circles(rng, count)
    ang = rng.uniform(0, 2 * PI, size=count)
    p = np.stack([xc + factor * r * np.cos(ang), yc + factor * r * np.sin(ang)], 1)
    q = p + np.stack([np.cos(ang + PI / 2), np.sin(ang + PI / 2)], 1)
    sp, ep = rng.uniform(-10, 10, size=[count, 1]), rng.uniform(-10, 10, size=[count, 1])
    return xc, yc, r, p, p + sp * (q - p), p + ep * (q - p)


def test_circle_one_intersection_tangent():
    rng = np.random.default_rng(2)
    xc, yc, r, p, start, end = _tangent_like(rng, 100, 1.0)
    plus, minus = geom.raw_line_circle_intersect(start[:, 0], start[:, 1], end[:, 0], end[:, 1],
                                                 xc, yc, r, 1e-6)
    assert bool(plus["valid"].all()) and bool(minus["valid"].all())
    sep = torch.sqrt((plus["x"] - minus["x"]) ** 2 + (plus["y"] - minus["y"]) ** 2)
    assert float(sep.max()) < 1e-6
    d = torch.sqrt((plus["x"] - torch.tensor(p[:, 0])) ** 2 + (plus["y"] - torch.tensor(p[:, 1])) ** 2)
    assert float(d.max()) < 1e-6


def test_circle_zero_intersections():
    rng = np.random.default_rng(3)
    xc, yc, r, p, start, end = _tangent_like(rng, 100, 2.0)
    plus, minus = geom.raw_line_circle_intersect(start[:, 0], start[:, 1], end[:, 0], end[:, 1],
                                                 xc, yc, r, 1e-10)
    assert not bool(plus["valid"].any()) and not bool(minus["valid"].any())


_GRID = [0.0, 0.00001, 1.0 / 4.0, 0.99999, 1.0, -0.00001, -1.0 / 4.0, -0.99999, -1.0]


def _in_interval(start, stop, count):
    if stop < start:
        stop = stop + 2 * PI
    a = np.linspace(start, stop, count)
    return np.where(a > PI, a - 2 * PI, a)


def _outside_interval(start, stop, count):
    if stop == start:
        start = start + 2 * PI
    return _in_interval(stop, start, count + 2)[1:-1]


@pytest.mark.parametrize("start", _GRID)
@pytest.mark.parametrize("stop", _GRID)
def test_angle_in_interval_grid(start, stop, count=11):
    start, stop = start * PI, stop * PI
    inc = _in_interval(start, stop, count)
    exc = _outside_interval(start, stop, count)
    if start == -PI and stop == PI:
        exc = np.zeros([0])
    elif start == PI and stop == -PI:
        inc = np.zeros([0])
    assert bool(geom.angle_in_interval(inc, start, stop).all())
    assert not bool(geom.angle_in_interval(exc, start, stop).any())
