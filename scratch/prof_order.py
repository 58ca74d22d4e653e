"""Runs the order / permute / restore entries a few times at one size (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import scene_util
from tensorflowraytrace_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
scene = scene_util.lens_scene(n, k_front=6, k_back=4, seed=3)
rays = torch.tensor(scene["rays"], dtype=torch.float32, device="cuda")
ids = torch.randperm(n, device="cuda").int()
counts = torch.zeros(8 * 4, dtype=torch.int32, device="cuda")
counts[2 * 8 + 1] = n; counts[3 * 8 + 1] = n
tab = torch.randn(2, n, dtype=torch.float64, device="cuda")
for _ in range(20):
    perm = ops.ray_order(rays)
    ops.permute_rays(rays, perm)
    ops.gather_rows(tab, perm)
    ops.restore_plan(ids, counts, 3, 1, perm, n)
torch.cuda.synchronize()
