"""
Differential fuzz of the coherent-ray path (k_intersect_beam) on scaled and shifted scenes: random
soups (no coherence: cuts, single rays, left-over wavefronts) and lens scenes (coherent wavefronts
proper) with several epsilons, ray-state dtypes, ray orders, new-ray lengths and pass counts, with
and without the grouped-kernel launch behind the beam kernel -- every ray set must equal the
natural-order trace bit for bit (which tests/test_gpu_stress.py pins against the oracle).
scratch/fuzz_coherent.py / fuzz_lens.py run the same over many more cases.
"""
import numpy as np
import pytest
import torch

import scene_util
import test_gpu_stress as st
from test_gpu_trace3d import _gpu_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CLASSES = ("finished", "active", "dead", "stopped", "unfinished")


def _same(out, ref, tag):
    assert np.array_equal(out["counts"], ref["counts"]), tag
    for cls in CLASSES:
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (tag, cls)
        assert torch.equal(out[cls], ref[cls]), (tag, cls)


@pytest.mark.parametrize("seed", [2, 7, 16, 21, 25, 35, 52, 58])
def test_scaled_and_shifted_soups(seed):
    from tensorflowraytrace_amd import ops, _lib
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    sc0 = st._soup(seed)
    if sc0["rays"].shape[1] < 64:
        pytest.skip("fewer rays than a wavefront")
    rng = np.random.default_rng(seed)
    scale = float(10.0 ** rng.uniform(-3, 3))
    shift = torch.tensor(rng.uniform(-1, 1, 3) * scale * float(10.0 ** rng.uniform(0, 2)))
    fv = (sc0["P"] * scale + shift.repeat(3)).to(DEV)
    rays = (sc0["rays"] * scale + shift.repeat(2).reshape(6, 1)).to(DEV)
    eps = [(1e-10, 1e-10, 1e-10), (1e-10 * scale ** 3, 1e-3, 1e-7), (1e-10, 0.2, -0.01)][seed % 3]
    base = dict(n_in=sc0["n_in"].to(DEV), n_out=sc0["n_out"].to(DEV))
    L = sc0["L"] * scale
    for dtype in (torch.float64, torch.float32):
        r = rays.to(dtype)
        plain = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), **base)
        plain.eps = eps
        ref = ops.trace3d(r, fv, plain, max_passes=4, flags=flags, new_ray_length=L)
        g = torch.Generator(device="cpu").manual_seed(seed)
        orders = {"hilbert": ops.ray_order(r),
                  "random": torch.randperm(r.shape[1], generator=g).int().to(DEV)}
        for name, order in orders.items():
            for only in (False, True):
                args = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv),
                                       coherent_rays=True, **base)
                args.eps = eps
                args.coherent_only = only
                raw = ops.trace3d(r[:, order.long()].contiguous(), fv, args, max_passes=4, flags=flags,
                                  new_ray_length=L)
                _same(ops.restore_order(raw, order), ref, (seed, dtype, name, only))


@pytest.mark.parametrize("case", range(8))
def test_scaled_and_shifted_lens_scenes(case):
    from tensorflowraytrace_amd import ops, _lib
    flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
    rng = np.random.default_rng(1000 + case)
    n_rays = int(rng.choice([3000, 9000, 20000, 45000]))
    kf, kb = int(rng.integers(3, 28)), int(rng.integers(3, 12))
    scene = scene_util.lens_scene(n_rays, k_front=kf, k_back=kb)
    dtype = torch.float32 if case % 2 else torch.float64
    src, fv, sc, _ = _gpu_scene(scene, dtype, cluster="group")
    scale = float(10.0 ** rng.uniform(-2, 2))
    shift = torch.tensor(rng.uniform(-1, 1, 3) * scale * float(10.0 ** rng.uniform(0, 1.5)), device=fv.device)
    fv = fv.detach() * scale + shift.repeat(3)
    src = (src.double() * scale + shift.repeat(2).reshape(6, 1)).to(dtype)
    eps = [(1e-10, 1e-10, 1e-10), (1e-12, 1e-4, 1e-8), (1e-10, 0.1, -0.02)][case % 3]
    eps = (eps[0] * scale ** 3, eps[1], eps[2])
    L = float(rng.choice([1.0, 0.01, 100.0])) * scale
    passes = int(rng.integers(2, 6))

    def args_for(order=None, coherent=False, only=False):
        a = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                            n_table=sc.n_table if order is None else sc.n_table[:, order.long()].contiguous(),
                            cluster_order=ops.cluster_order(fv), coherent_rays=coherent)
        a.eps = eps
        a.coherent_only = only
        return a

    ref = ops.trace3d(src, fv, args_for(), max_passes=passes, flags=flags, new_ray_length=L)
    assert int(ref["counts"][:, :4].sum()) > 0
    order = ops.ray_order(src)
    for only in (False, True):
        raw = ops.trace3d(src[:, order.long()].contiguous(), fv, args_for(order, True, only),
                          max_passes=passes, flags=flags, new_ray_length=L)
        _same(ops.restore_order(raw, order), ref, (case, only))
