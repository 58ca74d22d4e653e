"""Per-section shader-clock sums of k_intersect_group per pass on the bench scene (needs a
-DTFRT_GROUP_TIMING build: TFRT_LIB_PATH=scratch/variants/lib_timing.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
names = ["prologue", "tile+barrier", "level0", "rounds", "l1 batches", "member+lists", "screen", "decide", "epilogue"]
eng.ray_trace(3); torch.cuda.synchronize()
prev = np.zeros(16)
h.tfrt_debug_group_time(buf)
for P in (1, 2, 3):
    eng.ray_trace(P); torch.cuda.synchronize()
    h.tfrt_debug_group_time(buf)
    tot = np.array([buf[i] for i in range(16)], dtype=np.float64)
    cur = tot - (prev if P > 1 else 0); prev = tot
    waves = cur[15]; total = cur[:9].sum()
    print(f"pass {P}: waves {waves:.0f}, cycles/wave {total / waves:.0f}: " +
          ", ".join(f"{nm} {cur[i] / waves:.0f} ({100 * cur[i] / total:.0f}%)" for i, nm in enumerate(names)), flush=True)
