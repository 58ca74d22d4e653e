"""Timed-region overhead of the fused graph step: t(K) for K = 1..100 steps between synchronisations."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
eng, system, params = bench.build_scene(1_000_000, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6,
                              grad_clip=1e-3, fused="auto", graph="auto")
opt.suppress_warnings = True
for _ in range(20): opt.single_step(None)
torch.cuda.synchronize()
for K in (1, 2, 5, 10, 20, 50, 100, 20, 1):
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); time.sleep(0.002 if rep % 2 else 0.0)
        t0 = time.perf_counter()
        for _ in range(K): opt.single_step(None)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"K={K:4d}: {best*1e3:8.3f} ms total, {best/K*1e3:.4f} ms/step", flush=True)
