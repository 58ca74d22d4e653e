"""Host profile of the cfg2 step (system.update() + eng.ray_trace(5), 100k rays x 974 faces)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
eng, system, params = bench.build_scene(100_000, 9, 9, torch.float32)
def step():
    system.update()
    eng.ray_trace(5)
for _ in range(20): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t) / 200 * 1e3:.3f} ms")
t = time.perf_counter()
for _ in range(200): system.update()
torch.cuda.synchronize()
print(f"update alone {(time.perf_counter() - t) / 200 * 1e3:.3f} ms")
with torch.no_grad():
    t = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    print(f"step under no_grad {(time.perf_counter() - t) / 200 * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=70))
