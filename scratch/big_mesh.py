"""One-off: a 98k-face lens (k=128) x 200k rays; grouped vs all-pairs equality and timing."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
scene = scene_util.lens_scene(200_000, k_front=128, k_back=64)
outs = {}
for mode in ("group", False):
    src, fv, sc, _ = _gpu_scene(scene, torch.float32, cluster=mode)
    fv = fv.detach()
    for _ in range(2): out = ops.trace3d(src, fv, sc, max_passes=3, flags=_lib.COMPILE_FINISHED | _lib.COMPILE_DEAD)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): out = ops.trace3d(src, fv, sc, max_passes=3, flags=_lib.COMPILE_FINISHED | _lib.COMPILE_DEAD)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"mode={mode} M={fv.shape[0]} N=200000: {dt*1e3:.3f} ms fwd, {out['n_tests']/dt:.3e} pairs/s, finished {out['finished'].shape[1]}", flush=True)
    outs[mode] = out
a, b = outs["group"], outs[False]
print("identical:", torch.equal(a["finished"], b["finished"]) and torch.equal(a["finished_face"], b["finished_face"]) and torch.equal(a["dead"], b["dead"]))
