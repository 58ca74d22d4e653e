"""Times k_intersect3d alone (pass 0 of cfg4) through the tfrt_intersect3d seam."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = scene_util.lens_scene(N, k_front=41, k_back=9)
src, fv, sc, _ = _gpu_scene(scene, torch.float32)
fv = fv.detach()
lib = _lib.lib()
for _ in range(2): out = ops.intersect3d(src, fv)
torch.cuda.synchronize()
lib.tfrt_profile_enable(1)
for _ in range(8): out = ops.intersect3d(src, fv)
torch.cuda.synchronize()
buf = (ctypes.c_float * 64)(); n = lib.tfrt_profile_read(buf, 64)
ms = sorted(buf[i] for i in range(n))
tests = N * fv.shape[0]
print(f"{os.environ.get('TFRT_LIB_PATH','default').split('_')[-1]:>12s}  median {ms[n//2]:.3f} ms  min {ms[0]:.3f}  -> {tests/ms[n//2]/1e9:.2f}e12 tests/s   valid {int(out[3].sum())}")
