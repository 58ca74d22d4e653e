import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
from torch.profiler import profile, ProfilerActivity
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(12): opt.single_step(None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(10): opt.single_step(None)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=60))
