"""Duration of the first pass's k_intersect_beam launch (HIP events around eng.ray_trace(1) are too coarse:
run under rocprofv3 --kernel-trace --stats).  Usage: pass1_time.py RAYS"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
N = int(sys.argv[1])
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
eng.coherent = True
for _ in range(30):
    eng.ray_trace(1)
torch.cuda.synchronize()
