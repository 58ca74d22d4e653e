"""Funnel counters of k_intersect_group per pass on the cfg5a scene (stats build)."""
import sys, os, ctypes, importlib.util
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from tensorflowraytrace_amd import _lib
sys.argv = [sys.argv[0], "1000000"]
src = open(os.path.join(os.path.dirname(__file__), "perf_cfg5a.py")).read().split("ref = None")[0]
exec(src)
eng, system = build(torch.float32)
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
names = ["-", "(ray,super) pairs", "queued clusters", "member hits", "past screen", "-"]
prev = np.zeros(8)
for P in range(1, 7):
    torch.cuda.synchronize(); h.tfrt_debug_group_stats(buf)
    eng.ray_trace(P); torch.cuda.synchronize()
    h.tfrt_debug_group_stats(buf)
    tot = np.array([buf[i] for i in range(8)], dtype=np.float64)
    cur = tot - prev; prev = tot
    n_in = int(eng.last_trace["counts"][P - 1][:4].sum())
    print(f"pass {P}: rays {n_in}: " + ", ".join(f"{nm} {cur[i]/max(n_in,1):.2f}/ray" for i, nm in enumerate(names) if nm != "-"), flush=True)
