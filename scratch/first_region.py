"""bench.py's headline sequence (W warm-up steps, then K timed steps between synchronisations), the timed
region repeated: is the first one slower?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
W, K = int(sys.argv[1]), int(sys.argv[2])
IDLE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # seconds of idle chip before every region
eng, system, params = bench.build_scene(1_000_000, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6,
                              grad_clip=1e-3, fused="auto", graph="auto")
opt.suppress_warnings = True
for _ in range(W): opt.single_step(None)
fs = opt._fused_step
for rep in range(6):
    t_ = int(fs.tests_total.item()); torch.cuda.synchronize()
    if IDLE > 0.0 and rep >= 3:
        time.sleep(IDLE)
    t0 = time.perf_counter()
    for _ in range(K): opt.single_step(None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"region {rep}: {dt/K*1e3:.4f} ms/step  (replays so far {fs.graph_replays})", flush=True)
