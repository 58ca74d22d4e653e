"""N fused steps of the bench workload (for rocprofv3).  Usage: prof_step.py RAYS MODE STEPS [random]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]); mode = sys.argv[2]; K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
RANDOM = len(sys.argv) > 4 and sys.argv[4] == "random"
eng, system, params = bench.build_scene(N, 41, 9, torch.float32, random_rays=RANDOM)
if os.environ.get("TFRT_COHERENT"): eng.coherent = {"0": False, "1": True}.get(os.environ["TFRT_COHERENT"], "auto")
erf = bench.make_rowwise_error_function() if mode == "rowwise" else bench.make_error_function()
opt = optimizer.SGD_Optimizer(eng, params, erf, trace_depth=3, learning_rate=1e-6,
                              grad_clip=1e-3, fused=False if mode == "generic" else "auto",
                              graph="auto" if mode in ("graph", "rowwise") else False)
opt.suppress_warnings = True
for _ in range(10): opt.single_step(None)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(K): opt.single_step(None)
torch.cuda.synchronize()
print(f"N={N} {mode}: {(time.perf_counter() - t) / K * 1e3:.3f} ms/step")
