"""Differential fuzz on lens scenes (coherent wavefronts proper): scaled / shifted scenes, several
resolutions and ray counts, epsilons; sorted trace (with and without the grouped-kernel launch) against
the natural-order one, bit for bit.  fuzz_lens.py N_CASES"""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = 0
for case in range(int(sys.argv[1])):
    rng = np.random.default_rng(1000 + case)
    n_rays = int(rng.choice([3000, 9000, 20000, 45000, 130000]))
    kf, kb = int(rng.integers(3, 28)), int(rng.integers(3, 12))
    scene = scene_util.lens_scene(n_rays, k_front=kf, k_back=kb)
    dtype = torch.float32 if case % 2 else torch.float64
    src, fv, sc, _ = _gpu_scene(scene, dtype, cluster="group")
    fv = fv.detach()
    scale = float(10.0 ** rng.uniform(-2, 2))
    shift = torch.tensor(rng.uniform(-1, 1, 3) * scale * float(10.0 ** rng.uniform(0, 1.5)), device=fv.device)
    fv = fv * scale + shift.repeat(3)
    src = (src.double() * scale + shift.repeat(2).reshape(6, 1)).to(dtype)
    eps = [(1e-10, 1e-10, 1e-10), (1e-12, 1e-4, 1e-8), (1e-10, 0.1, -0.02)][case % 3]
    eps = (eps[0] * scale ** 3, eps[1], eps[2])
    L = float(rng.choice([1.0, 0.01, 100.0])) * scale
    passes = int(rng.integers(2, 6))
    def args_for(order=None, coherent=False, only=False):
        a = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                            n_table=sc.n_table if order is None else sc.n_table[:, order.long()].contiguous(),
                            cluster_order=ops.cluster_order(fv), coherent_rays=coherent)
        a.eps = eps; a.coherent_only = only
        return a
    ref = ops.trace3d(src, fv, args_for(), max_passes=passes, flags=flags, new_ray_length=L)
    order = ops.ray_order(src)
    for only in (False, True):
        raw = ops.trace3d(src[:, order.long()].contiguous(), fv, args_for(order, True, only), max_passes=passes,
                          flags=flags, new_ray_length=L)
        out = ops.restore_order(raw, order)
        ok = np.array_equal(out["counts"], ref["counts"])
        for cls in ("finished", "active", "dead", "stopped", "unfinished"):
            ok = ok and torch.equal(out[cls + "_id"], ref[cls + "_id"]) and torch.equal(out[cls], ref[cls])
        if not ok:
            bad += 1
            print("MISMATCH case", case, n_rays, kf, kb, dtype, "only", only, "scale %.3g" % scale, eps, "left", raw["left_over"], flush=True)
    print("case", case, n_rays, kf, kb, str(dtype)[6:], "scale %.2g L %.2g passes %d" % (scale, L, passes), "left over", raw["left_over"], "finished", int(ref["counts"][:, 1].sum()), flush=True)
print("mismatches:", bad)
