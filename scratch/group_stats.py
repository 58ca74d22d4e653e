"""Funnel counters of k_intersect_group per pass on the bench scene (needs the -DTFRT_GROUP_STATS
build: TFRT_LIB_PATH=scratch/libtfrt_stats.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
names = ["level-0 tests", "(ray,super) pairs", "queued clusters", "member hits", "past screen", "decisions hit"]
prev = np.zeros(8)
for P in (1, 2, 3):
    torch.cuda.synchronize(); h.tfrt_debug_group_stats(buf)
    eng.ray_trace(P); torch.cuda.synchronize()
    h.tfrt_debug_group_stats(buf)
    tot = np.array([buf[i] for i in range(8)], dtype=np.float64)
    cur = tot - prev; prev = tot
    counts = eng.last_trace["counts"]
    n_in = int(counts[P - 1][:4].sum())
    print(f"pass {P}: rays {n_in}: " + ", ".join(f"{nm} {cur[i]/n_in:.2f}/ray" for i, nm in enumerate(names)), flush=True)
