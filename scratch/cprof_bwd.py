import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
def timeit(tag, k=200):
    for _ in range(10): opt.single_step(None)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(k): opt.single_step(None)
    torch.cuda.synchronize(); print(tag, (time.perf_counter()-t)/k*1e3, "ms/step", flush=True)
timeit("multithreaded autograd")
torch.autograd.set_multithreading_enabled(False)
timeit("single-thread autograd")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): opt.single_step(None)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(45)
