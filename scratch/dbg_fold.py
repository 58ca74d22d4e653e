import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import test_gpu_fused_step as T
from tensorflowraytrace_amd import fused_step
orig = fused_step.FusedStep._enqueue_gradient
def wrapped(self):
    out = orig(self)
    eng = self.opt.engine
    st = self._state
    print("enqueue: folded", self.folded_backward, "perm", eng._trace_perm is not None, "M", st["M"], "N", st["N"], "capturing", torch.cuda.is_current_stream_capturing())
    return out
fused_step.FusedStep._enqueue_gradient = wrapped
opt, eng, system, lens, *_rest, acc = T._make(20000, "graph", k=6, ray_dtype=torch.float64)
T._run(opt, None, 8)
fs = opt._fused_step
print("final", fs.folded_backward, fs.capture_error, fs.graph_replays, eng.coherent, getattr(eng, "_order_cache", None) is not None)
