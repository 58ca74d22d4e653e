"""Differential fuzz of round 5's trace paths against the all-pairs trace, bit for bit, on random soups
(tests/test_gpu_stress._soup) with scaled / shifted coordinates and unusual epsilons, 2..6 passes:
  * in-place trace over the Hilbert order and over a random order, restored (ops.restore_order) and
    numbered by the caller (perm= : compacted through tfrt_scene3d.ray_slot, no restore),
  * natural-order hierarchy (the grouped walk with level 1's behind / beyond bounds).
fuzz_inplace.py FIRST_SEED N_SEEDS"""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, torch
import test_gpu_stress as st
from tensorflowraytrace_amd import ops, _lib
DEV = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
CLASSES = ("finished", "active", "dead", "stopped", "unfinished")
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0


def same(out, ref):
    ok = np.array_equal(out["counts"], ref["counts"])
    for cls in CLASSES:
        ok = ok and torch.equal(out[cls + "_id"], ref[cls + "_id"]) and torch.equal(out[cls], ref[cls])
        if cls != "unfinished":
            ok = ok and torch.equal(out[cls + "_face"], ref[cls + "_face"])
    return ok


for seed in range(first, first + count):
    sc0 = st._soup(seed)
    rng = np.random.default_rng(seed)
    scale = float(10.0 ** rng.uniform(-3, 3))
    shift = torch.tensor(rng.uniform(-1, 1, 3) * scale * float(10.0 ** rng.uniform(0, 2)))
    fv = (sc0["P"] * scale + shift.repeat(3)).to(DEV)
    rays = (sc0["rays"] * scale + shift.repeat(2).reshape(6, 1)).to(DEV)
    if rays.shape[1] < 64:
        continue
    eps = [(1e-10, 1e-10, 1e-10), (1e-10 * scale ** 3, 1e-3, 1e-7), (1e-10, 0.2, -0.01)][seed % 3]
    base = dict(n_in=sc0["n_in"].to(DEV), n_out=sc0["n_out"].to(DEV))
    L = sc0["L"] * scale
    passes = int(rng.integers(2, 7))
    dl = 0.5 * scale if seed % 2 else None
    cat = sc0["cat"].int().to(DEV)
    for dtype in (torch.float64, torch.float32):
        r = rays.to(dtype)
        plain = ops.Scene3DArgs(fv, cat, **base); plain.eps = eps
        ref = ops.trace3d(r, fv, plain, max_passes=passes, flags=flags, new_ray_length=L, dead_ray_length=dl)
        # natural-order hierarchy
        hier = ops.Scene3DArgs(fv, cat, cluster_order=ops.cluster_order(fv), **base); hier.eps = eps
        out = ops.trace3d(r, fv, hier, max_passes=passes, flags=flags, new_ray_length=L, dead_ray_length=dl)
        if not same(out, ref):
            bad += 1
            print("MISMATCH hierarchy seed", seed, dtype, "scale %.3g" % scale, eps, passes, flush=True)
        n = r.shape[1]
        g = torch.Generator(device="cpu").manual_seed(seed)
        orders = {"hilbert": ops.ray_order(r), "random": torch.randperm(n, generator=g).int().to(DEV)}
        for name, order in orders.items():
            for by_slot in (False, True):
                args = ops.Scene3DArgs(fv, cat, cluster_order=ops.cluster_order(fv), coherent_rays=True, **base)
                args.eps = eps
                args.coherent_only = args.in_place = True
                kw = dict(perm=order) if by_slot else {}
                raw = ops.trace3d(r[:, order.long()].contiguous(), fv, args, max_passes=passes, flags=flags,
                                  new_ray_length=L, dead_ray_length=dl, **kw)
                out = raw if by_slot else ops.restore_order(raw, order)
                if not same(out, ref):
                    bad += 1
                    print("MISMATCH in place seed", seed, dtype, name, "ray_slot" if by_slot else "restored",
                          "scale %.3g" % scale, eps, passes, flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done, mismatches so far", bad, flush=True)
print("mismatches:", bad)
