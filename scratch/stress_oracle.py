"""One-off stress: random soups with nasty scales / offsets / grazing rays, float64 ray state,
default trace mode vs the CPU oracle (dense float64 evaluation of the reference algorithm):
classes, order and hit faces must be identical (the float32 screen must never drop a hit).

Known benign mismatch: seed 35 (dead class, ray 453).  The scene generator puts a third of the
faces in one plane; two overlapping coplanar faces (32 and 57) then tie in ray_u up to the last
bit, the GPU and the oracle enter that pass with rays that differ in the last bit (their Snell
steps round differently) and pick different faces (scratch/debug_seed.py 35 shows both candidates;
the previous library gives the same result, and the all-pairs mode agrees with the hierarchy)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
from oracle import tracer
dev = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = 0; t0 = time.time(); cases = 0
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(5000 + seed)
    n_faces = int(rng.choice([64, 97, 300, 640]))
    n_rays = int(rng.choice([50, 700, 2500]))
    scale = 10 ** rng.uniform(-3, 3)
    offset = rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(0, 2.5) * (rng.random() < 0.5)
    centre = rng.uniform(-1, 1, (n_faces, 1, 3))
    size = 10 ** rng.uniform(-2.5, -0.2, (n_faces, 1, 1))
    tri = (centre + size * rng.standard_normal((n_faces, 3, 3))) * scale + offset
    if rng.random() < 0.5:
        tri[: n_faces // 3, :, 2] = offset[2] + 0.1 * scale
    P = torch.tensor(tri.reshape(n_faces, 9), dtype=torch.float64)
    cat = torch.zeros(n_faces, dtype=torch.int64)
    cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1; cat[int(0.9 * n_faces):] = 2
    n_in = torch.tensor(rng.uniform(1.0, 1.7, n_faces)); n_out = torch.tensor(rng.uniform(1.0, 1.7, n_faces))
    s = rng.uniform(-1.5, 1.5, (3, n_rays)) * scale + offset[:, None]
    d = rng.standard_normal((3, n_rays))
    if rng.random() < 0.5: d[2] *= 1e-3
    e = s + d * scale * 10 ** rng.uniform(-2, 0.5)
    rays = torch.tensor(np.concatenate([s, e]), dtype=torch.float64)
    fv = P.to(dev)
    sc = ops.Scene3DArgs(fv, cat.int().to(dev), n_in=n_in.to(dev), n_out=n_out.to(dev),
                         cluster_order=ops.cluster_order(fv))
    L = float(scale)
    out = ops.trace3d(rays.to(dev), fv, sc, max_passes=3, flags=flags, new_ray_length=L)
    def sub(mask):
        verts = P[mask].reshape(-1, 3)
        dd = tracer.faces_from_vertices(verts, torch.arange(verts.shape[0]).reshape(-1, 3))
        dd["n_in"] = n_in[mask]; dd["n_out"] = n_out[mask]
        return dd
    system = tracer.System(3, optical=sub(cat == 0), stop=sub(cat == 1), target=sub(cat == 2))
    src = {n: rays[i] for i, n in enumerate(("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"))}
    src["ray_id"] = torch.arange(n_rays, dtype=torch.float64)
    ref = tracer.ray_trace(system, src, max_iterations=3, inherit=("ray_id",), index_type="value",
                           new_ray_length=L, flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
    for cls in ("finished", "active", "stopped", "dead"):
        r = ref[cls]
        n_ref = r["x_start"].shape[0] if r else 0
        ok = out[cls].shape[1] == n_ref
        if ok and n_ref:
            ok = np.array_equal(out[cls + "_id"].cpu().numpy(), r["ray_id"].numpy().astype(np.int64))
            if ok:
                got = out[cls].cpu().numpy()
                want = np.stack([r[f].numpy() for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")])
                err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
                ok = err < 1e-9
        if not ok:
            if out[cls].shape[1] == n_ref and n_ref:
                ids_ok = np.array_equal(out[cls + "_id"].cpu().numpy(), r["ray_id"].numpy().astype(np.int64))
                print("  ids equal:", ids_ok, "rel err:", (err if ids_ok else None))
            bad += 1; print("MISMATCH seed", seed, cls, n_faces, n_rays, f"scale {scale:.3g} off {np.abs(offset).max():.3g}", out[cls].shape[1], n_ref, flush=True)
    cases += 1
print(f"{cases} cases, {bad} mismatches, {time.time()-t0:.1f} s")
