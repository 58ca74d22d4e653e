"""What the first optimiser steps of the bench workload do: wall time (with a synchronisation after each),
graph replays so far, in place or not."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
eng, system, params = bench.build_scene(1_000_000, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6,
                              grad_clip=1e-3, fused="auto", graph="auto")
opt.suppress_warnings = True
for k in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.single_step(None)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    fs = opt._fused_step
    print(f"step {k:2d}: host {1e3*(t1-t0):7.3f} ms, done {1e3*(t2-t0):7.3f} ms, replays {getattr(fs,'graph_replays',None)}, "
          f"eager {getattr(fs,'_eager_steps',None)}, in_place {getattr(fs,'in_place',None)}, capture_error {getattr(fs,'capture_error',None)}", flush=True)
