"""Stage clocks of k_trace_inplace on the bench scene (needs a -DTFRT_TUNING -DTFRT_TICKS build:
TFRT_LIB_PATH=scratch/variants_live/lib_ticks.so).  Usage: inplace_ticks.py [rays]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
eng.coherent = True
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
eng.ray_trace(3); eng.ray_trace(3); torch.cuda.synchronize()     # the second one runs in place
h.tfrt_debug_group_stats(buf)
for rep in range(2):
    eng.ray_trace(3); torch.cuda.synchronize()
    h.tfrt_debug_group_stats(buf)
    t = np.array([buf[i] for i in range(32)], dtype=np.float64)[16:]
    names = ["ray load", "bundle", "level 0", "level 1", "level 2", "(before faces)", "face_frame + order",
             "face walk", "decisions", "epilogue", "not narrow", "react + tape"]
    w = 15625 * N / 1e6
    print("ticks per wavefront (all passes): " + ", ".join(f"{n} {v / w:.0f}" for n, v in zip(names, t)) + f"; total {t[:12].sum() / w:.0f}")
