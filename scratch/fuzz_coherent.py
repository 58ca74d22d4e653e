"""Differential fuzz of the coherent-ray path: random soups (tests/test_gpu_stress._soup) with scaled /
shifted coordinates, every ray order, with and without the grouped-kernel launch, against the
all-pairs result, bit for bit.  fuzz_coherent.py FIRST_SEED N_SEEDS"""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, torch
import test_gpu_stress as st
from tensorflowraytrace_amd import ops, _lib
DEV = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
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
    for dtype in (torch.float64, torch.float32):
        r = rays.to(dtype)
        plain = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), **base); plain.eps = eps
        ref = ops.trace3d(r, fv, plain, max_passes=4, flags=flags, new_ray_length=L)
        n = r.shape[1]
        g = torch.Generator(device="cpu").manual_seed(seed)
        orders = {"hilbert": ops.ray_order(r), "random": torch.randperm(n, generator=g).int().to(DEV)}
        for name, order in orders.items():
            for only in (False, True):
                args = ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv),
                                       coherent_rays=True, **base)
                args.eps = eps; args.coherent_only = only
                raw = ops.trace3d(r[:, order.long()].contiguous(), fv, args, max_passes=4, flags=flags,
                                  new_ray_length=L)
                out = ops.restore_order(raw, order)
                ok = np.array_equal(out["counts"], ref["counts"])
                for cls in ("finished", "active", "dead", "stopped", "unfinished"):
                    ok = ok and torch.equal(out[cls + "_id"], ref[cls + "_id"]) and torch.equal(out[cls], ref[cls])
                if not ok:
                    bad += 1
                    print("MISMATCH seed", seed, dtype, name, only, "scale %.3g" % scale, eps, flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done, mismatches so far", bad, flush=True)
print("mismatches:", bad)
