import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
N = 125_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
for _ in range(5): system.update(); eng.ray_trace(3)
torch.cuda.synchronize()
def t(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("system.update host us:", t(system.update))
print("clear_ray_history host us:", t(eng.clear_ray_history))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): system.update()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
