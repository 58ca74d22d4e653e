"""Differential fuzz of the trace modes: all-pairs filter vs sphere hierarchy (natural order) on lens
scenes and soups with unusual epsilons, bit for bit.  fuzz_modes.py N_CASES"""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, torch
import scene_util
import test_gpu_stress as st
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = 0
EPS = [(1e-10, 0.3, 1e-10), (1e-10, 0.05, -0.05), (1e-8, 1.0, 1e-6), (1e-10, 1e-10, 1e-10)]
def same(a, b):
    ok = np.array_equal(a["counts"], b["counts"])
    for cls in ("finished", "active", "dead", "stopped", "unfinished"):
        ok = ok and torch.equal(a[cls + "_id"], b[cls + "_id"]) and torch.equal(a[cls], b[cls])
    return ok
for case in range(int(sys.argv[1])):
    rng = np.random.default_rng(500 + case)
    eps = EPS[case % len(EPS)]
    if case % 2 == 0:
        n_rays = int(rng.choice([3000, 20000, 60000]))
        kf, kb = int(rng.integers(3, 24)), int(rng.integers(3, 10))
        scene = scene_util.lens_scene(n_rays, k_front=kf, k_back=kb)
        src, fv, sc, _ = _gpu_scene(scene, torch.float64, cluster="group")
        fv = fv.detach()
        def args(cluster):
            a = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out, n_table=sc.n_table,
                                cluster_order=ops.cluster_order(fv) if cluster else None)
            a.eps = eps
            return a
        L, tag = 1.0, f"lens {n_rays} {kf} {kb}"
    else:
        sc0 = st._soup(case)
        fv, src = sc0["P"].to("cuda:0"), sc0["rays"].to("cuda:0")
        def args(cluster):
            a = ops.Scene3DArgs(fv, sc0["cat"].int().to("cuda:0"), n_in=sc0["n_in"].to("cuda:0"),
                                n_out=sc0["n_out"].to("cuda:0"),
                                cluster_order=ops.cluster_order(fv) if cluster else None)
            a.eps = eps
            return a
        L, tag = sc0["L"], f"soup {case}"
    ref = ops.trace3d(src, fv, args(False), max_passes=4, flags=flags, new_ray_length=L)
    out = ops.trace3d(src, fv, args(True), max_passes=4, flags=flags, new_ray_length=L)
    if not same(out, ref):
        bad += 1
        print("MISMATCH", tag, eps, flush=True)
print("cases", sys.argv[1], "mismatches:", bad)
