import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
from torch.profiler import profile, ProfilerActivity
eng, system, params = bench.build_scene(1_000_000, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(12): opt.single_step(None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(5): opt.single_step(None)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6, group_by_input_shape=True)
rows = [e for e in ka if e.self_device_time_total > 0 and not e.key.startswith(("void tfrt", "tfrt::"))]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:28]:
    st = [s for s in e.stack if "tensorflowraytrace_amd" in s or "bench.py" in s][:2]
    print(f"{e.self_device_time_total/5:8.1f} us/step x{e.count/5:.1f} {e.key[:38]:38s} {str(e.input_shapes)[:40]:40s} {' <- '.join(x.split('/')[-1][:60] for x in st)}")
