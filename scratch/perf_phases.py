import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(5): opt.single_step(None)
torch.cuda.synchronize()
import collections
acc = collections.defaultdict(float)
K = 30
pc = time.perf_counter
for _ in range(K):
    t0 = pc(); eng.clear_ray_history(); system.update(); t1 = pc()
    eng.speculative_counts = True
    eng.ray_trace(3); t2 = pc()
    err = bench.error_function(eng); es = err.sum(); t3 = pc()
    g = torch.autograd.grad(es, params, retain_graph=True); t4 = pc()
    ok = eng.verify_trace(); t5 = pc()
    proc = []
    for gi in g:
        gi = torch.where(torch.isfinite(gi), gi, torch.zeros_like(gi)); gi = gi * 1e-6; gi = torch.clamp(gi, -1e-3, 1e-3); proc.append(gi)
    opt.apply_gradients(proc); t6 = pc()
    for k, v in (("update", t1-t0), ("trace", t2-t1), ("error", t3-t2), ("grad", t4-t3), ("verify(wait)", t5-t4), ("opt", t6-t5)):
        acc[k] += v
torch.cuda.synchronize()
for k, v in acc.items(): print(f"{k:14s} {v/K*1e3:.3f} ms")
print("total cpu", sum(acc.values())/K*1e3)
