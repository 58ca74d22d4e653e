"""RCCL sanity on one GPU: process group of size 1, the collectives the sharded step uses."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
buf = torch.arange(5000, dtype=torch.float64, device="cuda")
dist.all_reduce(buf, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(buf.sum()), float(t))
dist.destroy_process_group()
