"""Unsynchronised blocks of 10 steps: ms/step, speculation misses and miss-rate estimate."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for blk in range(8):
    m0 = opt.speculation_misses
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(10):
        opt.single_step(None)
    torch.cuda.synchronize()
    print(f"block {blk}: {(time.perf_counter()-t)*100:.3f} ms/step  misses {opt.speculation_misses-m0}  rate {opt._miss_rate:.2f}", flush=True)
