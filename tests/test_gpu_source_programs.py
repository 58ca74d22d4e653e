"""Sources made on the device (csrc/tfrt_source.hip: tfrt_points_generate, tfrt_source3d_generate,
tfrt_epoch_advance) behind the reference's source / distribution classes: the Random*
distributions (tfrt/distributions.py:1375-1393, 1586-1598, 1751-1775, 1814-1850),
BasePointTransformation (:2014-2120) and the Aperature / Point / Angular sources
(tfrt/sources.py:464-1095), and the optimiser step over a source that is re-drawn every step
(dev/hexalens.py:36-48, tfrt/optimizer.py:217).  The reference's generator is TensorFlow's and
unseeded: parity is the distribution (moments, supports), the assembly of the rays (exact), and
that nothing else about a trace depends on how the rays were made."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
PI = math.pi


def _dist():
    import tfrt.distributions as d
    return d


def test_random_circle_is_uniform_on_the_disc_and_redrawn_by_update():
    d = _dist()
    d.seed(11)
    n = 400000
    c = d.RandomUniformCircle(n, 0.7)
    assert c.__dict__.get("_device_active")
    p1 = c.points
    assert p1.shape == (n, 2) and p1.is_cuda and p1.dtype == torch.float64
    assert c.points is p1                                   # the same draw until the next update
    r, th = c.polar_ranks[:, 0], c.polar_ranks[:, 1]
    assert float(r.min()) >= 0 and float(r.max()) < 1 and float(th.min()) >= 0 and float(th.max()) < 2 * PI
    assert abs(float((r ** 2).mean()) - 0.5) < 4 * math.sqrt(1 / 12 / n)       # r^2 ~ U(0, 1)
    assert abs(float(th.mean()) - PI) < 4 * math.sqrt((2 * PI) ** 2 / 12 / n)
    assert abs(float((r ** 2 * torch.cos(th)).mean())) < 4 / math.sqrt(n)       # r and theta independent
    assert torch.allclose(p1, 0.7 * c.ranks, rtol=0, atol=1e-15)
    c.update()
    p2 = c.points
    assert p2 is not p1 and float((p2 - p1).abs().max()) > 0.1
    # theta wedge (distributions.py:1396-1447)
    w = d.RandomUniformCircle(10000, 1.0, theta_start=0.0, theta_end=PI / 6)
    th = w.polar_ranks[:, 1]
    assert float(th.min()) >= 0 and float(th.max()) < PI / 6


def test_random_square_and_spheres():
    d = _dist()
    d.seed(3)
    q = d.RandomUniformSquare(0.5, 300, 0.25, 200)
    p = q.points
    assert p.shape == (60000, 2)
    assert float(p[:, 0].abs().max()) <= 0.5 and float(p[:, 1].abs().max()) <= 0.25
    assert abs(float(p[:, 0].var()) - 0.5 ** 2 / 3) < 0.01 and abs(float(p[:, 1].var()) - 0.25 ** 2 / 3) < 0.003
    assert torch.allclose(q.ranks, p / 0.5)
    n = 200000
    for cls, power in ((d.RandomUniformSphere, 1), (d.RandomLambertianSphere, 2)):
        s = cls(0.6, n, radius=2.0)
        v = s.points
        assert v.shape == (n, 3)
        assert torch.allclose(v.norm(dim=1), torch.full((n,), 2.0, dtype=torch.float64, device=DEV), atol=1e-13)
        c = (v[:, 0] / 2.0) ** power                        # cos(phi)^power ~ U(cos(a)^power, 1)
        lo = math.cos(0.6) ** power
        assert float(c.min()) >= lo - 1e-12 and float(c.max()) <= 1.0
        assert abs(float(c.mean()) - (1 + lo) / 2) < 4 * (1 - lo) / math.sqrt(12 * n)
        assert torch.allclose(s.ranks[:, 0], torch.acos(v[:, 0] / 2.0), atol=1e-7)
        assert s.angles is s.points


def test_transformation_is_part_of_the_program_from_the_next_update_on():
    d = _dist()
    d.seed(5)
    c = d.RandomUniformCircle(5000, 0.2)
    q = d.quaternion_from_euler((0.3, -0.2, 0.5))
    d.BasePointTransformation(c, rotation=q, translation=(-10.0, 1.0, 2.0), scale=(1.0, 2.0, 0.5))
    assert c.points.shape == (5000, 2)                      # like the reference: a post-update handle
    c.update()
    p = c.points
    assert p.shape == (5000, 3)
    plane = 0.2 * c.ranks                                   # the same draw, before the transformation
    lifted = torch.cat([torch.zeros_like(plane[:, :1]), plane], dim=1) * torch.tensor(
        [1.0, 2.0, 0.5], dtype=torch.float64, device=DEV)
    want = d.rotate_vector_by_quaternion(q, lifted) + torch.tensor([-10.0, 1.0, 2.0], dtype=torch.float64, device=DEV)
    assert torch.allclose(p, want, rtol=0, atol=1e-13)


def _aperture(n, wavelengths=(575.0,), static_end=False):
    import tfrt.sources as sources
    d = _dist()
    a = d.RandomUniformCircle(n, 0.2)
    d.BasePointTransformation(a, translation=(-10, 0, 0))
    b = (d.StaticUniformCircle if static_end else d.RandomUniformCircle)(n, 0.9)
    d.BasePointTransformation(b)
    src = sources.AperatureSource(3, a, b, list(wavelengths), dense=False,
                                  extra_fields={"object_coords": ("start_point", a, "points"),
                                                "ap_ranks": ("end_point", b, "polar_ranks")})
    return src, a, b


@pytest.mark.parametrize("static_end", [False, True])
def test_aperture_source_fields_blocks_orders_and_shards_are_one_draw(static_end):
    import tfrt.sources as sources
    d = _dist()
    d.seed(7)
    n = 30000
    src, a, b = _aperture(n, static_end=static_end)
    src.update()
    rs = src._fields
    assert isinstance(rs, sources.DeviceRaySet)
    assert set(rs.keys()) == {"x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength",
                              "object_coords", "ap_ranks"}
    start = torch.stack([src["x_start"], src["y_start"], src["z_start"]], dim=1)
    end = torch.stack([src["x_end"], src["y_end"], src["z_end"]], dim=1)
    assert torch.equal(start, a.points) and torch.equal(end, b.points)      # sources.py:1016-1020
    assert torch.equal(src["object_coords"], a.points) and torch.equal(src["ap_ranks"], b.polar_ranks)
    assert src["wavelength"].shape == (n,) and float(src["wavelength"][0]) == 575.0
    fields = torch.cat([start, end], dim=1).t()
    for dt in (torch.float32, torch.float64, torch.float16):
        blk = rs.ray_block(dt)
        if dt == torch.float16:       # (float64 -> float16 may round through float32: one ulp)
            assert bool(((blk.double() - fields).abs() <= 2.0 ** -10 * fields.abs() + 1e-7).all())
        else:
            assert torch.equal(blk, fields.to(dt))
        assert rs.ray_block(dt) is blk                       # persistent
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(n, generator=g).int().to(DEV)
    pv = rs.permuted(perm)
    assert torch.equal(pv.ray_block(torch.float32), fields.float()[:, perm.long()])
    assert torch.equal(pv["y_end"], src["y_end"][perm.long()])
    assert torch.equal(pv["object_coords"], a.points[perm.long()])
    assert torch.equal(pv["ap_ranks"], b.polar_ranks[perm.long()])
    sh = rs.shard(1000, 8000)
    assert torch.equal(sh.ray_block(torch.float64), fields[:, 1000:8000])
    assert torch.equal(sh["object_coords"], a.points[1000:8000])
    old = fields.clone()
    blk32 = rs.ray_block(torch.float32)
    src.update()
    assert src._fields.ray_block(torch.float32) is blk32     # in place
    new = torch.stack([src[f] for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")])
    assert torch.equal(blk32, new.float())
    assert float((new[1] - old[1]).abs().max()) > 0.01
    if static_end:
        assert torch.equal(new[3:], old[3:])


@pytest.mark.parametrize("n", [30_000, 300_000])
def test_order_of_a_program_is_a_permutation_as_compact_as_the_order_of_its_rays(n):
    """tfrt_source3d_order makes the keys straight from the program (float32 evaluation, extents of
    the key grid from 1024 sampled rays): not the same permutation as tfrt_ray_order over the
    generated block, but one whose 64-ray wavefronts cover patches of the aperture just as small."""
    from tensorflowraytrace_amd import ops
    d = _dist()
    d.seed(11)
    src, a, b = _aperture(n)
    src.update()
    rs = src._fields
    perm = rs.order()
    assert perm.dtype == torch.int32 and perm.shape == (n,)
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=perm.device))
    block = rs.ray_block(torch.float32)
    ref = ops.ray_order(block, None, axis=src.axis_hint())

    def spread(p):      # mean diagonal of the end-point bounding box of a wavefront's 64 rays
        yz = block[4:6, p.long()][:, :n // 64 * 64].reshape(2, -1, 64)
        ext = yz.max(dim=2)[0] - yz.min(dim=2)[0]
        return float(ext.pow(2).sum(dim=0).sqrt().mean())

    assert spread(perm) <= 1.1 * spread(ref)
    g = torch.Generator().manual_seed(0)
    assert spread(perm) < 0.2 * spread(torch.randperm(n, generator=g).to(perm.device))
    # a new draw: a new order from the same launch sequence
    src.update()
    perm2 = src._fields.order()
    assert not torch.equal(perm, perm2)
    assert torch.equal(torch.sort(perm2.long())[0], torch.arange(n, device=perm.device))


@pytest.mark.parametrize("n", [1, 50, 3_000, 30_000, 300_000, 1_000_000])
def test_the_faster_order_of_a_program_equals_the_sorted_one_up_to_ties(n):
    """tfrt_source3d_order_cells (``order(stable=False)``): one scatter by the key's high digit, then
    every bucket sorted by the low digit in LDS.  Same keys, so the same order wherever the keys
    differ: positions may only differ inside runs of equal keys (about one ray per key cell, and
    few rays share one)."""
    d = _dist()
    d.seed(17)
    src, a, b = _aperture(n)
    src.update()
    rs = src._fields
    stable = rs.order()
    perm = rs.order(stable=False)
    assert perm.dtype == torch.int32 and perm.shape == (n,)
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=perm.device))
    same = perm == stable
    # where they differ, the rays are the same set within a short window (a run of equal keys)
    if not bool(same.all()):
        bad = torch.nonzero(~same).flatten()
        assert bad.numel() < 0.6 * n          # (most cells hold one ray)
        W = 64
        for j in bad[:: max(1, bad.numel() // 50)].tolist():
            lo, hi = max(0, j - W), min(n, j + W)
            assert int(perm[j]) in set(stable[lo:hi].tolist())
    if n >= 30_000:
        block = rs.ray_block(torch.float32)

        def spread(p):
            yz = block[4:6, p.long()][:, :n // 64 * 64].reshape(2, -1, 64)
            ext = yz.max(dim=2)[0] - yz.min(dim=2)[0]
            return float(ext.pow(2).sum(dim=0).sqrt().mean())
        assert spread(perm) <= 1.02 * spread(stable)
    # again into the same buffer: still a permutation
    perm2 = rs.order(stable=False, out=perm)
    assert perm2 is perm
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=perm.device))


def test_point_and_angular_sources_assemble_like_the_torch_path():
    import tfrt.sources as sources
    d = _dist()
    d.seed(9)
    n = 20000
    ang = d.RandomLambertianSphere(0.4, n)
    ps = sources.PointSource(3, (1.0, -2.0, 0.5), (1.0, 1.0, 0.2), ang, [500.0], dense=False,
                             start_on_center=False, ray_length=2.5)
    ps.update()
    assert isinstance(ps._fields, sources.DeviceRaySet)
    v = ps._rotate_angles(ang.points)
    c = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64, device=DEV)
    end = torch.stack([ps["x_start"], ps["y_start"], ps["z_start"]], 1)      # swapped
    start = torch.stack([ps["x_end"], ps["y_end"], ps["z_end"]], 1)
    assert torch.allclose(start, c.expand(n, 3), atol=1e-14)
    assert torch.allclose(end, c + 2.5 * v, atol=1e-13)
    base = d.RandomUniformSquare(0.3, 200, 0.1, 100)
    ang2 = d.RandomUniformSphere(0.2, n)
    an = sources.AngularSource(3, (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), ang2, base, [500.0], dense=False)
    an.update()
    assert isinstance(an._fields, sources.DeviceRaySet)
    s0 = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64, device=DEV) + an._rotate_points(base.points)
    e0 = s0 + an._rotate_angles(ang2.points)
    assert torch.allclose(torch.stack([an["x_start"], an["y_start"], an["z_start"]], 1), s0, atol=1e-13)
    assert torch.allclose(torch.stack([an["x_end"], an["y_end"], an["z_end"]], 1), e0, atol=1e-13)


def test_order_of_point_and_angular_programs_and_of_tiny_sources():
    """tfrt_source3d_order on the other source kinds: a point source (directions without a common
    axis: the octahedral map), an angular source (a frame from the mean direction), and sources of
    fewer rays than the 256 samples the frame is made from -- always a permutation; where there
    are wavefronts, their end points lie closer together than those of a random order."""
    import tfrt.sources as sources
    d = _dist()
    d.seed(13)

    def check(src, n):
        src.update()
        rs = src._fields
        perm = rs.order()
        assert perm.dtype == torch.int32 and perm.shape == (n,)
        assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=perm.device))
        if n < 64 * 32:
            return
        block = rs.ray_block(torch.float32)

        def spread(p):
            e = block[3:6, p.long()][:, :n // 64 * 64].reshape(3, -1, 64)
            return float((e.max(dim=2)[0] - e.min(dim=2)[0]).pow(2).sum(dim=0).sqrt().mean())
        g = torch.Generator().manual_seed(0)
        assert spread(perm) < 0.35 * spread(torch.randperm(n, generator=g).to(perm.device))

    n = 40000
    ang = d.RandomUniformSphere(PI, n)                       # every direction: no common axis
    check(sources.PointSource(3, (1.0, -2.0, 0.5), (1.0, 0.0, 0.0), ang, [500.0], dense=False,
                              ray_length=2.5), n)
    base = d.RandomUniformSquare(0.3, 200, 0.1, 200)         # (200 x 200 = n points)
    ang2 = d.RandomUniformSphere(0.2, n)
    check(sources.AngularSource(3, (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), ang2, base, [500.0], dense=False), n)
    for m in (1, 63, 100, 257):
        src, a, b = _aperture(m)
        check(src, m)


def test_trace_of_a_device_made_source_equals_the_trace_of_its_rays_as_plain_tensors():
    """ray_trace() over a re-drawn source runs ordered (tfrt_ray_order every trace) and restored;
    the same rays handed over as a ManualSource and traced in natural order give every ray set
    bit for bit."""
    import tfrt.sources as sources
    from test_gpu_engine import _build_lens
    d = _dist()
    d.seed(21)
    eng, system, lens, target, source = _build_lens(20000, k=6, ray_dtype=torch.float32, random_rays=True,
                                                    compile_dead_rays=True, compile_stopped_rays=True)
    for _ in range(2):
        system.update()
        eng.ray_trace(4)
        assert eng._trace_perm is not None                  # ordered
        got = {c: {f: getattr(eng, c + "_rays")[f].clone() for f in ("x_start", "y_end", "z_end", "object_coords")}
               for c in ("finished", "active", "dead") if bool(getattr(eng, c + "_rays"))}
        manual = sources.ManualSource(3)
        for f in source.keys():
            manual[f] = source[f].clone()
        eng2, system2, *_ = _build_lens(20000, k=6, ray_dtype=torch.float32, coherent=False,
                                        compile_dead_rays=True, compile_stopped_rays=True)
        system2.sources = [manual]
        system2.update()
        eng2.ray_trace(4)
        assert eng2._trace_perm is None
        for c, fields in got.items():
            for f, v in fields.items():
                assert torch.equal(v, getattr(eng2, c + "_rays")[f]), (c, f)


def _optimizers(n_rays, seed, ray_dtype=torch.float64):
    import tfrt.optimizer as optimizer
    from test_gpu_engine import _build_lens
    d = _dist()
    out = {}
    for mode in ("generic", "graph"):
        d.seed(seed)
        kw = dict(coherent=False) if mode == "generic" else {}
        eng, system, lens, target, source = _build_lens(n_rays, k=5, ray_dtype=ray_dtype, random_rays=True, **kw)
        erf = optimizer.GoalError(("y_end", "z_end"), lambda src: -src["object_coords"][:, 1:],
                                  rowwise=(mode == "graph"))
        opt = optimizer.SGD_Optimizer(eng, lens.parameters, erf, 3, learning_rate=3e-4, grad_clip=1e9,
                                      fused=False if mode == "generic" else "auto",
                                      graph="auto" if mode == "graph" else False, speculative=False)
        opt.suppress_warnings = True
        out[mode] = (opt, eng, lens, source)
    return out


def test_fused_step_over_a_redrawn_source_equals_the_generic_natural_order_step():
    """dev/hexalens.py's loop: the source is re-drawn at every step.  The fused step draws it in
    place, orders it on the device and replays one launch graph; the generic step traces the same
    draws (same seed, same streams) in natural order through torch autograd."""
    runs = _optimizers(12000, seed=33)
    steps = 9
    errs, params = {}, {}
    for mode, (opt, eng, lens, source) in runs.items():
        errs[mode] = [float(opt.single_step(None)) for _ in range(steps)]
        params[mode] = [p.detach().cpu().clone() for p in lens.parameters]
    fs = runs["graph"][0]._fused_step
    assert fs.capture_error is None, fs.capture_error
    assert fs.graph_replays >= steps - 5
    assert runs["graph"][1]._trace_perm is not None
    assert len(set(errs["generic"])) == steps               # a new draw every step
    np.testing.assert_allclose(errs["graph"], errs["generic"], rtol=1e-10, atol=0)
    for a, b in zip(params["graph"], params["generic"]):
        assert float((a - b).abs().max()) <= 1e-10 * float(b.abs().max())
    # the ray sets of the last (replayed) step, cut lazily, belong to the last draw
    opt, eng, lens, source = runs["graph"]
    fin = eng.finished_rays
    ids = eng.last_trace["finished_id"].long()
    assert torch.equal(fin["object_coords"], source["object_coords"][ids])


def test_a_source_with_a_differentiable_center_keeps_the_torch_path():
    """A device program bakes center / central_angle / the transformation in as numbers: when one of
    them requires grad the source must stay on the differentiable torch path, or d error / d center
    would silently be zero (the reference's tape reaches them: tfrt/sources.py:464-1095 are plain
    tensor ops)."""
    import tfrt.sources as sources
    d = _dist()
    d.seed(5)
    n = 5000
    center = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64, device=DEV, requires_grad=True)
    ps = sources.PointSource(3, center, (1.0, 1.0, 0.2), d.RandomUniformSphere(0.3, n), [500.0],
                             dense=False)
    ps.update()
    assert not isinstance(ps._fields, sources.DeviceRaySet)
    err = (ps["x_end"] ** 2 + ps["y_start"] * 3.0).sum()
    g, = torch.autograd.grad(err, [center])
    assert float(g.abs().min()) > 0.0 or float(g.abs().max()) > 0.0
    assert abs(float(g[1]) - 3.0 * n) < 1e-6 * n            # d (3 sum y_start) / d center_y
    # the same source without a gradient request is made on the device
    ps2 = sources.PointSource(3, center.detach(), (1.0, 1.0, 0.2), d.RandomUniformSphere(0.3, n),
                              [500.0], dense=False)
    ps2.update()
    assert isinstance(ps2._fields, sources.DeviceRaySet)
    # a transformation that requires grad keeps its distribution on the torch path too
    shift = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64, device=DEV, requires_grad=True)
    pts = d.RandomUniformCircle(n, 0.5)
    d.BasePointTransformation(pts, translation=shift)
    pts.update()
    pts.update()
    assert not pts.__dict__.get("_device_active")
    gs, = torch.autograd.grad(pts.points.sum(), [shift])
    assert torch.allclose(gs, torch.full_like(gs, float(n)))
