"""cfg5b: 2-D scene with 64 arcs + 256 segments, N rays; forward + backward timing."""
import sys, os, time, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
from oracle import tracer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = "cuda:0"
rng = np.random.default_rng(0)
t = lambda a, dt=torch.float64: torch.tensor(np.asarray(a), dtype=dt, device=dev)
# 64 lenslet arcs in a row, a 255-segment wavy mirror above, a target wall
na = 64
xc = np.linspace(-16, 16, na)
arc_geo = np.stack([xc, np.full(na, 3.0), np.full(na, -math.pi + 0.3), np.full(na, -0.3), np.full(na, 0.6)], 1)
xs = np.linspace(-18, 18, 256)
ys = 6.0 + 0.3 * np.sin(xs)
seg_geo = np.stack([xs[:-1], ys[:-1], xs[1:], ys[1:]], 1)
wall = np.array([[20.0, -1, 20.0, 9.0]])
seg_all = np.concatenate([seg_geo, wall])
i32 = torch.int32
seg = dict(geo=t(seg_all).requires_grad_(True), cat=t([0]*255+[2], i32), mat_in=t([2]*255+[0], i32), mat_out=t([0]*256, i32), n_in=None, n_out=None)
arc = dict(geo=t(arc_geo).requires_grad_(True), cat=t([0]*na, i32), mat_in=t([1]*na, i32), mat_out=t([0]*na, i32), n_in=None, n_out=None)
ang = rng.uniform(0.3*math.pi, 0.7*math.pi, N); x0 = rng.uniform(-15, 15, N)
rays = torch.tensor(np.stack([x0, np.zeros(N), x0+np.cos(ang), np.sin(ang)]), dtype=torch.float32, device=dev)
wl = torch.full((N,), 550.0, dtype=torch.float64)
n_table = torch.stack([tracer.MATERIALS[m](wl) for m in ("vacuum","acrylic","reflective")]).to(dev)
scene = ops.Scene2DArgs(seg, arc, n_table, True, False)
for bwd in (False, True):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.trace2d(rays, scene, 4, flags=_lib.COMPILE_FINISHED)
        if bwd:
            loss = (out["finished"][3].double() ** 2).sum()
            g = torch.autograd.grad(loss, [seg["geo"], arc["geo"]])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(("fwd+bwd" if bwd else "fwd"), f"N={N}: {dt*1e3:.2f} ms, {out['n_tests']/dt:.3e} tests/s, counts {out['counts'][:, :4].tolist()}")
