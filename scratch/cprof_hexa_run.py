import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "examples"))
import torch, hexalens
t = time.perf_counter(); errors, s = hexalens.run(ray_count=20000, steps=20, verbose=False); print("first run (20 steps)", time.perf_counter() - t)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); errors, s = hexalens.run(ray_count=20000, steps=60, verbose=False); dt = time.perf_counter() - t
pr.disable()
print("second run: 60 steps in", dt, "s ->", dt / 60 * 1e3, "ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
