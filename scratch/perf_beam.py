"""Step time with and without the coherent visiting order.  Usage: perf_beam.py [rays ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 125_000]
for N in sizes:
    for coherent in (False, True):
        eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
        eng.coherent = coherent
        opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
        opt.suppress_warnings = True
        for _ in range(10): opt.single_step(None)
        torch.cuda.synchronize(); K = 50; t = time.perf_counter()
        for _ in range(K): e = opt.single_step(None)
        torch.cuda.synchronize()
        fs = opt._fused_step
        print(f"N={N} coherent={coherent}: {(time.perf_counter() - t) / K * 1e3:.3f} ms/step  err {float(e):.6e} graph_replays {fs.graph_replays} capture_error {fs.capture_error}", flush=True)
        del eng, system, params, opt
