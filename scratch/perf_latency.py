"""Launch time of the pass-0 intersect kernel against ray count (tfrt_profile events)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
lib = _lib.lib()
for N in [int(a) for a in sys.argv[1:]]:
    eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
    for _ in range(3): eng.ray_trace(3)
    torch.cuda.synchronize(); lib.tfrt_profile_enable(1)
    for _ in range(20): eng.ray_trace(3)
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 4096)(); n = lib.tfrt_profile_read(buf, 4096); lib.tfrt_profile_enable(0)
    ms = np.array([buf[i] for i in range(n)]).reshape(-1, 3)
    print(f"N={N:8d} per-pass us {np.round(np.median(ms,0)*1e3,1)}", flush=True)
