"""Host time per fused step over a long run (does the enqueue rate hold?).  host_rate.py RAYS STEPS"""
import sys, os, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch, bench
import tfrt.optimizer as optimizer
N, K = int(sys.argv[1]), int(sys.argv[2])
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(10): opt.single_step(None)
torch.cuda.synchronize()
for chunk in range(K // 500):
    t = time.perf_counter()
    for _ in range(500): opt.single_step(None)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"steps {chunk*500:5d}..: host enqueue {t_host/500*1e3:.3f} ms/step, with drain {t_all/500*1e3:.3f} ms/step", flush=True)
