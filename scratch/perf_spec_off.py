import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
for N in (125_000, 250_000, 500_000):
    for spec in (True, False):
        eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
        opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
        opt.suppress_warnings = True; opt.speculative = spec
        for _ in range(20): opt.single_step(None)
        torch.cuda.synchronize(); t = time.perf_counter(); m0 = opt.speculation_misses
        for _ in range(100): opt.single_step(None)
        torch.cuda.synchronize()
        print(f"N={N} speculative={spec}: {(time.perf_counter()-t)*10:.3f} ms/step, misses {opt.speculation_misses-m0}/100", flush=True)
