"""
Differential fuzz of round 5's trace paths against the ALL-PAIRS trace (every ray against every face,
the arithmetic tests/test_gpu_stress.py pins to the oracle), bit for bit, on random triangle soups with
scaled and shifted coordinates, unusual epsilons (a negative ray_start_epsilion among them), 2-6 passes and
a dead-ray length:

  * the in-place trace (tfrt_scene3d.in_place: all passes in one launch) over the Hilbert order and over a
    random order of the rays, restored afterwards (ops.restore_order) and numbered by the caller
    (``perm=``: compacted through tfrt_scene3d.ray_slot, nothing restored),
  * the natural-order sphere hierarchy, whose level 1 drops clusters wholly behind the ray's start and
    wholly beyond its nearest hit so far.

A soup has no coherence: wavefronts are cut down to single rays, faces overlap and are coplanar in
places, mirrors and total internal reflection occur, all four ray classes are populated.
scratch/fuzz_inplace.py runs the same over thousands of seeds (3,150 without a mismatch).
"""
import numpy as np
import pytest
import torch

import test_gpu_stress as st

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CLASSES = ("finished", "active", "dead", "stopped", "unfinished")


def _same(out, ref, tag):
    assert np.array_equal(out["counts"], ref["counts"]), tag
    for cls in CLASSES:
        assert torch.equal(out[cls + "_id"], ref[cls + "_id"]), (tag, cls)
        assert torch.equal(out[cls], ref[cls]), (tag, cls)
        if cls != "unfinished":
            assert torch.equal(out[cls + "_face"], ref[cls + "_face"]), (tag, cls)


@pytest.mark.parametrize("seed", [1001, 1002, 1005, 1006, 1007, 1008, 1010, 1011, 1013, 1015, 1022, 1028])
def test_in_place_and_hierarchy_traces_equal_the_all_pairs_trace(seed):
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
    passes = int(rng.integers(2, 7))
    dl = 0.5 * scale if seed % 2 else None
    cat = sc0["cat"].int().to(DEV)
    kw = dict(max_passes=passes, flags=flags, new_ray_length=L, dead_ray_length=dl)
    for dtype in (torch.float64, torch.float32):
        r = rays.to(dtype)
        plain = ops.Scene3DArgs(fv, cat, **base)
        plain.eps = eps
        ref = ops.trace3d(r, fv, plain, **kw)
        hier = ops.Scene3DArgs(fv, cat, cluster_order=ops.cluster_order(fv), **base)
        hier.eps = eps
        _same(ops.trace3d(r, fv, hier, **kw), ref, (seed, dtype, "hierarchy"))
        g = torch.Generator(device="cpu").manual_seed(seed)
        orders = {"hilbert": ops.ray_order(r),
                  "random": torch.randperm(r.shape[1], generator=g).int().to(DEV)}
        for name, order in orders.items():
            for by_slot in (False, True):
                args = ops.Scene3DArgs(fv, cat, cluster_order=ops.cluster_order(fv), coherent_rays=True,
                                       **base)
                args.eps = eps
                args.coherent_only = args.in_place = True
                raw = ops.trace3d(r[:, order.long()].contiguous(), fv, args,
                                  **(dict(kw, perm=order) if by_slot else kw))
                out = raw if by_slot else ops.restore_order(raw, order)
                _same(out, ref, (seed, dtype, name, "ray_slot" if by_slot else "restored"))
