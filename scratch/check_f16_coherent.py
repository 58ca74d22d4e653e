"""float16 ray state: sorted trace against the natural-order one, bit for bit (lens scene, 4 passes)."""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = 0
for n_rays, kf in ((20000, 12), (70000, 20)):
    scene = scene_util.lens_scene(n_rays, k_front=kf, k_back=6)
    src, fv, sc, _ = _gpu_scene(scene, torch.float16, cluster="group")
    fv = fv.detach()
    ref = ops.trace3d(src, fv, sc, max_passes=4, flags=flags)
    order = ops.ray_order(src.float())
    for only in (False, True):
        a = ops.Scene3DArgs(fv, sc.catagory, mat_in=sc.mat_in, mat_out=sc.mat_out,
                            n_table=sc.n_table[:, order.long()].contiguous(), cluster_order=sc.cluster_order,
                            coherent_rays=True)
        a.coherent_only = only
        raw = ops.trace3d(src[:, order.long()].contiguous(), fv, a, max_passes=4, flags=flags)
        out = ops.restore_order(raw, order)
        ok = np.array_equal(out["counts"], ref["counts"])
        for cls in ("finished", "active", "dead", "stopped", "unfinished"):
            ok = ok and torch.equal(out[cls + "_id"], ref[cls + "_id"]) and torch.equal(out[cls], ref[cls])
        print(n_rays, kf, "only", only, "ok", ok, "left over", raw["left_over"], "counts", ref["counts"][:, :4].sum(0), flush=True)
        bad += 0 if ok else 1
print("mismatches", bad)
