"""Counters of k_intersect_beam per pass on the bench scene (needs a -DTFRT_TUNING build:
TFRT_LIB_PATH=scratch/variants_live/lib_tune.so).  Usage: beam_stats.py [rays]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
RANDOM = len(sys.argv) > 2 and sys.argv[2] == "random"
eng, system, params = bench.build_scene(N, 41, 9, torch.float32, random_rays=RANDOM)
eng.coherent = True
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
last = np.zeros(32)
for P in (1, 2, 3):
    torch.cuda.synchronize(); h.tfrt_debug_group_stats(buf)
    eng.ray_trace(P); torch.cuda.synchronize()
    h.tfrt_debug_group_stats(buf)
    tot = np.array([buf[i] for i in range(32)], dtype=np.float64)
    whole = tot                             # all P passes of this trace (reading clears the counters)
    cur = whole - last; last = whole        # its last pass
    w = max(cur[8], 1)
    ticks_only = cur[8] == 0     # (-DTFRT_TICKS: the counters are off)
    if not ticks_only:
        print(f"pass {P}: wavefronts {cur[8]:.0f}, left over {cur[9:13].sum():.0f} (spread {cur[9]:.0f}, supers {cur[10]:.0f}, "
              f"clusters {cur[11]:.0f}, faces {cur[12]:.0f}); per wavefront: member spheres touched {cur[13]/w:.1f}, "
              f"faces past face_frame {cur[27]/w:.1f}, faces walked {cur[29]/w:.2f}, pairs queued {cur[14]/w:.1f}, "
              f"decision batches {cur[28]/w:.2f}, bundles tried {cur[30]/w:.2f}; group kernel: queued clusters {cur[2]:.0f}"
              + (f"; pairs by surface: front {cur[21]/w:.1f}, back {cur[22]/w:.1f}, target {cur[23]/w:.1f}" if cur[21:24].sum() > 0 else ""), flush=True)
    names = ["ray load", "bundle", "level 0", "level 1", "level 2", "(before faces)", "face_frame + order",
             "face walk", "decisions", "epilogue", "not narrow"]
    if cur[8] == 0:
        w = {1: 15625, 2: 13682, 3: 13682}[P] * (N / 1e6)   # (-DTFRT_TICKS: the counters are off)
    tk = cur[16:16 + len(names)] / w
    if ticks_only:
        print(f"pass {P}: shader-clock ticks per wavefront and stage: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, tk)) + f"; total {tk.sum():.0f}", flush=True)
    if cur[9:13].sum() == 1:
        q0 = int(cur[15])
        rec = eng._order_cache[1]
        print("left-over range starts at slot", q0)
        r = rec[q0 - 8:q0 + 16].cpu()
        idx = r.view(torch.int32)[:, 6]
        for k in range(r.shape[0]):
            print(q0 - 8 + k, int(idx[k]), [round(float(v), 5) for v in r[k, :6]])
    if not ticks_only and cur[4] > 0:
        print(f"   decisions: pairs decided {cur[4]/w:.1f} per wavefront, of them valid hits {cur[5]/w:.1f}", flush=True)
