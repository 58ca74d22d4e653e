"""ms/step of the bench optimiser step at several per-GPU ray counts (strong-scaling shards)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
ACC = sys.argv[1] if len(sys.argv) > 1 else 'auto'
ACC = {'acc': True, 'none': False}.get(ACC, ACC)
SIZES = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1_000_000, 500_000, 250_000, 125_000)
for N in SIZES:
    eng, system, params = bench.build_scene(N, 41, 9, torch.float32, accelerate=ACC)
    opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
    opt.suppress_warnings = True
    for _ in range(5): opt.single_step(None)
    torch.cuda.synchronize()
    K = 40
    t = time.perf_counter()
    for _ in range(K): e = opt.single_step(None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"mode={ACC} N={N}: {dt*1e3:.3f} ms/step  err {float(e):.6g} misses {opt.speculation_misses}/{opt.iterations}", flush=True)
