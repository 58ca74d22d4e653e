"""Per-step wall times of the first 30 optimiser steps (synchronised each step)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
eng, system, params = bench.build_scene(1_000_000, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
ts = []
for i in range(30):
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.single_step(None)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(" ".join(f"{x:.2f}" for x in ts), "| misses", opt.speculation_misses)
