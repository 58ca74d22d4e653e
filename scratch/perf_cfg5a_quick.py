"""cfg5a (4M rays x 15,106 faces, ray_trace(8)) timing + per-launch intersect times."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import scene_configs as sc5
from tensorflowraytrace_amd import _lib
lib = _lib.lib()
eng, system, parts = sc5._build_5a(torch.float32, compile_all=False)
for _ in range(3): eng.ray_trace(sc5.PASSES_5A); eng.last_trace
torch.cuda.synchronize(); t = time.perf_counter()
K = 5
for _ in range(K): eng.ray_trace(sc5.PASSES_5A); eng.last_trace
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
lib.tfrt_profile_enable(1)
eng.ray_trace(sc5.PASSES_5A); eng.last_trace; torch.cuda.synchronize()
buf = (ctypes.c_float * 64)()
n = lib.tfrt_profile_read_kind(0, buf, 64)
lib.tfrt_profile_enable(0)
print(f"cfg5a {dt*1e3:7.2f} ms  ordered={eng._trace_perm is not None}  intersect launches ms: {[round(buf[i], 3) for i in range(max(n, 0))]}")
