"""Soak: thousands of fused optimiser steps (HIP-graph replays): memory stays flat, the error keeps falling,
parameters stay finite; then the same optimiser continues on the generic path and agrees."""
import sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch, bench
import tfrt.optimizer as optimizer
N, K = int(sys.argv[1]), int(sys.argv[2])
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-5, grad_clip=1e-3)
opt.suppress_warnings = True
errs, mem = [], []
t = time.perf_counter()
for k in range(K):
    e = opt.single_step(None)
    if k % (K // 10) == 0:
        torch.cuda.synchronize()
        errs.append(float(e)); mem.append(torch.cuda.memory_allocated() / 2**20)
torch.cuda.synchronize()
fs = opt._fused_step
print(f"N={N}: {K} steps in {time.perf_counter() - t:.2f} s, replays {fs.graph_replays}, capture_error {fs.capture_error}")
print("error every tenth:", [f"{x:.6g}" for x in errs])
print("allocated MiB:", [f"{m:.1f}" for m in mem])
print("finite:", all(bool(torch.isfinite(p).all()) for p in params), "finished rays", int(eng.finished_rays["x_start"].shape[0]))
assert max(mem[2:]) - min(mem[2:]) < 1.0, "memory grows"
assert errs[-1] < errs[0]
