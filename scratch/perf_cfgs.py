"""Timings of the other BASELINE configurations through the ops layer (default trace mode)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib

def run(tag, N, kf, kb, passes, dtype, cluster, K=8):
    scene = scene_util.lens_scene(N, k_front=kf, k_back=kb)
    src, fv0, sc, (p_f, p_b) = _gpu_scene(scene, dtype, cluster=cluster)
    dev = src.device
    tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt, device=dev)
    zf, zb, vec = tt(scene["zero_f"]), tt(scene["zero_b"]), tt(scene["vector"]).reshape(1, 3)
    ff, fb = tt(scene["faces_f"], torch.int32), tt(scene["faces_b"], torch.int32)
    fv_t, _ = ops.build_faces(tt(scene["target_verts"]), tt(scene["target_faces"], torch.int32))
    def faces():
        a, _ = ops.build_faces(zf + p_f.reshape(-1, 1) * vec, ff)
        b, _ = ops.build_faces(zb + p_b.reshape(-1, 1) * vec, fb)
        return torch.cat([a, b, fv_t])
    def step(bwd):
        fv = faces()
        out = ops.trace3d(src, fv, sc, max_passes=passes, flags=_lib.COMPILE_FINISHED)
        if bwd:
            fin = out["finished"]
            err = (fin[4].double() ** 2 + fin[5].double() ** 2).sum()
            torch.autograd.grad(err, [p_f, p_b])
        return out
    for bwd in (False, True):
        for _ in range(2): out = step(bwd)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(K): out = step(bwd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
        print(f"{tag:34s} {'fwd+bwd' if bwd else 'fwd    '} N={N} M={fv0.shape[0]} passes={passes} {str(dtype)[6:]:8s} "
              f"{'group' if cluster else 'all-pairs'}: {dt*1e3:7.3f} ms  {out['n_tests']/dt:.3e} pairs/s", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "cfg2"):
    for cl in ("group", False):
        run("cfg2/3: 100k rays x 974 faces", 100_000, 9, 9, 5, torch.float32, cl)
if which in ("all", "cfg5"):
    for dt in (torch.float32, torch.float16):
        run("cfg5a-like: 4M rays x 974 faces", 4_000_000, 9, 9, 3, dt, "group", K=4)
    run("4M rays x 10574 faces", 4_000_000, 41, 9, 3, torch.float32, "group", K=4)
