"""Optimiser step time of the bench workload by ray count and step mode (generic / fused eager /
fused + HIP graph).  Usage: perf_step2.py [rays ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import bench
import tfrt.optimizer as optimizer
sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 500_000, 250_000, 125_000]
print(f"{'rays':>9} {'generic':>9} {'fused':>9} {'graph':>9}   ms/step")
for N in sizes:
    row = []
    for mode in ("generic", "eager", "graph"):
        eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
        opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6,
                                      grad_clip=1e-3, fused=False if mode == "generic" else "auto",
                                      graph="auto" if mode == "graph" else False)
        opt.suppress_warnings = True
        for _ in range(10): opt.single_step(None)
        torch.cuda.synchronize()
        K = 50
        t = time.perf_counter()
        for _ in range(K): e = opt.single_step(None)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t) / K * 1e3)
        if mode == "graph":
            fs = opt._fused_step
            assert fs.capture_error is None, fs.capture_error
            assert fs.graph_replays >= K
        del eng, system, params, opt
        torch.cuda.empty_cache()
    print(f"{N:9d} {row[0]:9.3f} {row[1]:9.3f} {row[2]:9.3f}", flush=True)
