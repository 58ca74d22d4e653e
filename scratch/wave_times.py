"""Timeline of k_intersect_beam's wavefronts in the LAST pass of a P-pass trace (needs a
-DTFRT_TUNING -DTFRT_TICKS build).  Usage: wave_times.py RAYS PASSES"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]); P = int(sys.argv[2])
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
eng.coherent = True
h = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(3):
    eng.ray_trace(P); torch.cuda.synchronize()
t0 = np.zeros(65536, dtype=np.uint64); t1 = np.zeros(65536, dtype=np.uint64)
info = np.zeros(65536, dtype=np.uint64)
h.tfrt_debug_wave_times(t0.ctypes.data_as(ctypes.c_void_p), t1.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p))
att = (info >> np.uint64(48)).astype(np.int64); dec = ((info >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
fac = ((info >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64); cand = (info & np.uint64(0xFFFF)).astype(np.int64)
ok = t1 > 0
# earlier passes (more wavefronts) leave stale entries: keep the last cluster of start times
st = np.sort(t0[ok].astype(np.int64)); gaps = np.nonzero(np.diff(st) > 1000)[0]
if len(gaps): ok &= t0.astype(np.int64) > st[gaps[-1]]
a = t0[ok].astype(np.int64); b = t1[ok].astype(np.int64)
base = a.min(); a -= base; b -= base
life = (b - a) / 100.0    # us
print(f"N={N} pass {P}: {ok.sum()} wavefronts, launch span {b.max()/100:.1f} us; life us: mean {life.mean():.2f} p50 {np.percentile(life,50):.2f} "
      f"p90 {np.percentile(life,90):.2f} p99 {np.percentile(life,99):.2f} max {life.max():.2f}")
edges = np.linspace(0, b.max(), 21)
act = [int(((a <= (edges[i]+edges[i+1])/2) & (b > (edges[i]+edges[i+1])/2)).sum()) for i in range(20)]
print("  resident wavefronts at 20 instants:", act)
print(f"  per wavefront: bundles tried {att[ok].mean():.2f} (max {att[ok].max()}), decisions batches {dec[ok].mean():.2f} (max {dec[ok].max()}), faces screened {fac[ok].mean():.2f} (max {fac[ok].max()})")
print("  candidate faces histogram (<=8, <=16, <=32, <=64, <=128, more):", [int(((cand[ok] > a_) & (cand[ok] <= b_)).sum()) for a_, b_ in ((-1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 10**6))])
print("  faces screened histogram (<=4, <=8, <=16, <=32, <=64, more):", [int(((fac[ok] > a_) & (fac[ok] <= b_)).sum()) for a_, b_ in ((-1, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 10**6))])
lf = (t1.astype(np.int64) - t0.astype(np.int64)) / 100.0
for a_, b_ in ((-1, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 10**6)):
    m = ok & (fac > a_) & (fac <= b_)
    if m.any(): print(f"    faces {a_+1}..{b_}: mean life {lf[m].mean():.1f} us")
idx = np.nonzero(ok)[0]
slow = idx[np.argsort(-life)[:12]]
print("  slowest wavefronts (index: life us, start us, bundles, decision batches, faces):", [(int(i), round(float((int(t1[i])-int(t0[i]))/100.0),1), round(float((int(t0[i])-base)/100.0),1), int(att[i]), int(dec[i]), int(fac[i]), int(cand[i])) for i in slow])
for lo_, hi_ in ((1, 1), (2, 3), (4, 255)):
    m = ok & (att >= lo_) & (att <= hi_)
    if m.any(): print(f"  bundles {lo_}..{hi_}: {m.sum()} wavefronts, mean life {((t1[m].astype(np.int64)-t0[m].astype(np.int64))/100.0).mean():.1f} us")
late = idx[np.argsort(-b)[:8]]
print("  last to finish (index: life us, start us):", [(int(i), round(float((t1[i]-t0[i])/100.0),1), round(float((int(t0[i])-base)/100.0),1)) for i in late])
