"""Per-pass launch time of the intersect kernel in the bench step (tfrt_profile events)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, bench
from tensorflowraytrace_amd import _lib
import tfrt.optimizer as optimizer
MODE = sys.argv[1] if len(sys.argv) > 1 else 'auto'
MODE = {'acc': True, 'none': False}.get(MODE, MODE)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32, accelerate=MODE)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
ONE = os.environ.get("ONE_PASS") == "1"   # ablated libraries: only pass 1 is meaningful
step = (lambda: eng.ray_trace(1)) if ONE else (lambda: opt.single_step(None))
for _ in range(3): step()
torch.cuda.synchronize()
lib = _lib.lib(); lib.tfrt_profile_enable(1)
K = 10
for _ in range(K): step()
torch.cuda.synchronize()
buf = (ctypes.c_float * 4096)(); n = lib.tfrt_profile_read(buf, 4096)
ms = np.array([buf[i] for i in range(n)]).reshape(-1, 1 if ONE else 3)
tag = os.environ.get('TFRT_LIB_PATH', 'default').split('/')[-1]
print(f"{tag:28s} mode={MODE} N={N} per-pass ms {np.round(ms.mean(0), 3)} total {ms.mean(0).sum():.3f}", flush=True)
