"""One-off stress: many random soups (incl. nasty scales / offsets / grazing rays); every trace
mode must reproduce the all-pairs result bit for bit, and the all-pairs result must agree with
a brute-force float64 evaluation of the exact test on all pairs (first pass)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
dev = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = 0; t0 = time.time(); cases = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    rng = np.random.default_rng(1000 + seed)
    n_faces = int(rng.choice([64, 97, 300, 1500, 6000]))
    n_rays = int(rng.choice([7, 300, 5000, 30000]))
    scale = 10 ** rng.uniform(-3, 3)                     # overall scene scale
    offset = rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(0, 3) * (rng.random() < 0.5)
    centre = rng.uniform(-1, 1, (n_faces, 1, 3))
    size = 10 ** rng.uniform(-3.0, -0.2, (n_faces, 1, 1))
    tri = (centre + size * rng.standard_normal((n_faces, 3, 3))) * scale + offset
    if rng.random() < 0.5:                               # a coplanar sheet: grazing incidence
        tri[: n_faces // 3, :, 2] = offset[2] + 0.1 * scale
    tri[::41, 2] = tri[::41, 1]
    fv = torch.tensor(tri.reshape(n_faces, 9), dtype=torch.float64, device=dev)
    cat = torch.zeros(n_faces, dtype=torch.int32, device=dev)
    cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1; cat[int(0.9 * n_faces):] = 2
    n_in = torch.tensor(rng.uniform(1.0, 1.7, n_faces), device=dev)
    n_out = torch.tensor(rng.uniform(1.0, 1.7, n_faces), device=dev)
    s = rng.uniform(-1.5, 1.5, (3, n_rays)) * scale + offset[:, None]
    d = rng.standard_normal((3, n_rays))
    if rng.random() < 0.5: d[2] *= 1e-3                  # nearly in the sheet's plane
    e = s + d * scale * 10 ** rng.uniform(-2, 0.5)
    dtype = torch.float32 if rng.random() < 0.7 else torch.float64
    rays = torch.tensor(np.concatenate([s, e]), dtype=dtype, device=dev)
    def scene(mode):
        order = ops.cluster_order(fv) if mode else None
        return ops.Scene3DArgs(fv, cat, n_in=n_in, n_out=n_out, cluster_order=order, sort_rays=mode == "sort")
    ref = ops.trace3d(rays, fv, scene(False), max_passes=3, flags=flags, dead_ray_length=2.0 * scale)
    for mode in ("group", "sort"):
        out = ops.trace3d(rays, fv, scene(mode), max_passes=3, flags=flags, dead_ray_length=2.0 * scale)
        ok = np.array_equal(out["counts"], ref["counts"])
        for cls in ("finished", "active", "stopped", "dead"):
            ok = ok and torch.equal(out[cls + "_face"], ref[cls + "_face"]) and torch.equal(out[cls], ref[cls])
        if not ok:
            bad += 1; print("MISMATCH seed", seed, mode, n_faces, n_rays, scale, dtype, flush=True)
    # brute force, first pass: exact float64 test of every pair through the seam (all-pairs, no hierarchy)
    x, y, z, valid, ray_u, tu, tv, gi = ops.intersect3d(rays, fv)
    first = ops.trace3d(rays, fv, scene("group"), max_passes=1, flags=flags)
    n_hit = int(valid.sum()); n_dead = first["dead"].shape[1]
    if n_hit + n_dead != n_rays:
        bad += 1; print("COUNT MISMATCH seed", seed, n_hit, n_dead, n_rays, flush=True)
    cases += 1
print(f"{cases} cases, {bad} mismatches, {time.time()-t0:.1f} s")
