"""The sharded optimiser step through RCCL with a process group of size 1 (the code path of N > 1:
shard bounds, fused gradient buffer, nccl all-reduce, device-side count) against the plain step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, torch.distributed as dist
import bench
import tfrt.optimizer as optimizer
from tensorflowraytrace_amd import distributed as tdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29532")
torch.cuda.set_device(0)

def run(steps):
    eng, system, params = bench.build_scene(200_000, 41, 9, torch.float32)
    opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-5, grad_clip=1e-3)
    opt.suppress_warnings = True
    errs = [float(opt.single_step(None)) for _ in range(steps)]
    return errs, [p.detach().clone() for p in params]

plain_e, plain_p = run(5)
dist.init_process_group("nccl", rank=0, world_size=1)
tdist.is_distributed = lambda: True           # force the N > 1 path with one rank
dist_e, dist_p = run(5)
torch.cuda.synchronize()
print("errors plain", plain_e)
print("errors rccl ", dist_e)
for a, b in zip(plain_p, dist_p):
    print("max |param diff|", float((a - b).abs().max()))
dist.destroy_process_group()
