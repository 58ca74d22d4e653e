import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kf = int(sys.argv[2]) if len(sys.argv) > 2 else 41
dtype = torch.float32
scene = scene_util.lens_scene(N, k_front=kf, k_back=9)
src, fv, sc, (p_f, p_b) = _gpu_scene(scene, dtype)
M = fv.shape[0]
dev = src.device
tt = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt, device=dev)
zf, zb, vec = tt(scene["zero_f"]), tt(scene["zero_b"]), tt(scene["vector"]).reshape(1,3)
ff, fb = tt(scene["faces_f"], torch.int32), tt(scene["faces_b"], torch.int32)
fv_t, _ = ops.build_faces(tt(scene["target_verts"]), tt(scene["target_faces"], torch.int32))
def faces():
    a, _ = ops.build_faces(zf + p_f.reshape(-1,1)*vec, ff)
    b, _ = ops.build_faces(zb + p_b.reshape(-1,1)*vec, fb)
    return torch.cat([a, b, fv_t])
print("N", N, "M", M, flush=True)
def step(bwd):
    fv = faces()
    out = ops.trace3d(src, fv, sc, max_passes=3, flags=_lib.COMPILE_FINISHED)
    if bwd:
        fin = out["finished"]
        err = (fin[4].double() ** 2 + fin[5].double() ** 2).sum()
        g = torch.autograd.grad(err, [p_f, p_b])
    return out
for bwd in (False, True):
    out = step(bwd); torch.cuda.synchronize()
    t = time.time(); K = 5
    for _ in range(K): out = step(bwd)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    print("bwd" if bwd else "fwd", f"{dt*1e3:.2f} ms/step", f"{out['n_tests']/dt:.3e} tests/s", out["counts"][:, :4].tolist(), flush=True)
