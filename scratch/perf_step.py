"""Step-time breakdown of the bench workload at a given per-GPU ray count."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
ACC = len(sys.argv) > 2 and sys.argv[2] == 'acc'
eng, system, params = bench.build_scene(N, 41, 9, torch.float32, accelerate=ACC)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(3): opt.single_step(None)
torch.cuda.synchronize()
K = 20
t = time.perf_counter()
for _ in range(K): opt.single_step(None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"N={N}: {dt*1e3:.3f} ms/step  misses {opt.speculation_misses}/{opt.iterations}")
# host-only cost: time without waiting (async enqueue time)
def timed(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("update      %.3f ms" % timed(system.update))
print("ray_trace   %.3f ms" % timed(lambda: eng.ray_trace(3)))
def fwd_err():
    eng.ray_trace(3); return bench.error_function(eng).sum()
print("trace+err   %.3f ms" % timed(fwd_err))
def full_grad():
    system.update(); e = fwd_err(); torch.autograd.grad(e, params)
print("upd+trace+err+grad %.3f ms" % timed(full_grad))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): opt.single_step(None)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
