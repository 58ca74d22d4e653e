"""Host profile of system.update() on the cfg2 scene."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
eng, system, params = bench.build_scene(100_000, 9, 9, torch.float32)
for _ in range(50): system.update()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(500): system.update()
torch.cuda.synchronize(); print(f"update {(time.perf_counter() - t) / 500 * 1e3:.4f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(500): system.update()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
