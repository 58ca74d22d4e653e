"""Funnel counters of k_intersect_group over one fused optimiser step (3 passes), per source ray
(needs the -DTFRT_GROUP_STATS build: TFRT_LIB_PATH=scratch/variants/lib_stats.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
import tfrt.optimizer as optimizer
from tensorflowraytrace_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6,
                              grad_clip=1e-3, graph=False)
opt.suppress_warnings = True
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
names = ["level-0 tests", "(ray,super) pairs", "queued clusters", "member hits", "past screen", "decisions hit"]
for step in range(4):
    torch.cuda.synchronize(); h.tfrt_debug_group_stats(buf)
    opt.single_step(None); torch.cuda.synchronize()
    h.tfrt_debug_group_stats(buf)
    cur = np.array([buf[i] for i in range(8)], dtype=np.float64)
    print(f"step {step} (hints {eng.trace_hints}): " + ", ".join(f"{nm} {cur[i]/N:.2f}" for i, nm in enumerate(names)), flush=True)
