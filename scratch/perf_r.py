import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import scene_util
from test_gpu_trace3d import _gpu_scene
from tensorflowraytrace_amd import ops, _lib
N = int(sys.argv[1])
scene = scene_util.lens_scene(N, k_front=41, k_back=9)
src, fv, sc, _ = _gpu_scene(scene, torch.float32, cluster="group")
fv = fv.detach()
for _ in range(2): out = ops.trace3d(src, fv, sc, max_passes=3, flags=_lib.COMPILE_FINISHED)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): out = ops.trace3d(src, fv, sc, max_passes=3, flags=_lib.COMPILE_FINISHED)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"R={os.environ.get('TFRT_GROUP_RAYS_PER_LANE','auto')} N={N}: {dt*1e3:.3f} ms fwd")
