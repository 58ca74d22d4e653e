"""Times of the device ray order, the permutation and the order restoration (us per call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import scene_util
from tensorflowraytrace_amd import ops, _lib

def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def graph_time(f, reps=30):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); 
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            f()
    torch.cuda.synchronize()
    return timeit(g.replay, reps)

for n in (125000, 1000000, 4000000):
    scene = scene_util.lens_scene(n, k_front=6, k_back=4, seed=3)
    rays = torch.tensor(scene["rays"], dtype=torch.float32, device="cuda")
    perm = ops.ray_order(rays)
    t_order = timeit(lambda: ops.ray_order(rays))
    t_order_g = graph_time(lambda: ops.ray_order(rays))
    t_perm = timeit(lambda: ops.permute_rays(rays, perm))
    t_perm_g = graph_time(lambda: ops.permute_rays(rays, perm))
    t_torch = timeit(lambda: rays[:, perm.long()].contiguous())
    tab = torch.randn(2, n, dtype=torch.float64, device="cuda")
    t_rows = timeit(lambda: ops.gather_rows(tab, perm))
    ids = torch.randperm(n, device="cuda").int()
    counts = torch.zeros(8 * 4, dtype=torch.int32, device="cuda")
    counts[2 * 8 + 1] = n; counts[3 * 8 + 1] = n      # all finished in pass 2 (P = 3)
    t_plan = timeit(lambda: ops.restore_plan(ids, counts, 3, 1, perm, n))
    print(f"n={n}: order {t_order:.1f} (graph {t_order_g:.1f}) permute {t_perm:.1f} (graph {t_perm_g:.1f}; torch index {t_torch:.1f}) "
          f"gather 2 f64 rows {t_rows:.1f} restore plan {t_plan:.1f} us", flush=True)
