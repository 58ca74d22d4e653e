"""Which torch ops (with their Python call sites) run in one steady-state fused optimiser step."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
from torch.profiler import profile, ProfilerActivity
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1e-6, grad_clip=1e-3, graph=False)
opt.suppress_warnings = True
for _ in range(8): opt.single_step(None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2): opt.single_step(None)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.device_type.name == "CPU" and ev.name.startswith("aten::") and ev.cuda_time_total > 0 if hasattr(ev, "cuda_time_total") else False:
        pass
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.key.startswith("aten::") and getattr(ev, "device_time_total", 0) > 0:
        stack = [s for s in ev.stack if "tensorflowraytrace_amd" in s or "bench.py" in s][:3]
        print(f"{ev.key:28s} x{ev.count/2:4.1f} dev_us/step {ev.device_time_total/2:7.1f}  {' | '.join(s.split('/')[-1] for s in stack)}")
