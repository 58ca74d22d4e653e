import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import faulthandler; faulthandler.enable()
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
stage = sys.argv[1]; dev = "cuda:0"
p = torch.randn(1000, dtype=torch.float64, device=dev, requires_grad=True)
w = torch.randn(1000, dtype=torch.float64, device=dev)
go = torch.randn(3000, dtype=torch.float64, device=dev)
def f():
    y = torch.cat([p * w, p + 1.0, p * p])
    return y
# warm-up
for _ in range(3):
    if stage.startswith("st"):
        with torch.autograd.set_multithreading_enabled(False):
            torch.autograd.grad([f()], [p], grad_outputs=[go])
    else:
        torch.autograd.grad([f()], [p], grad_outputs=[go])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if stage == "st_torch":
    with torch.cuda.graph(g):
        with torch.autograd.set_multithreading_enabled(False):
            gr = torch.autograd.grad([f()], [p], grad_outputs=[go])
elif stage == "mt_torch":
    with torch.cuda.graph(g):
        gr = torch.autograd.grad([f()], [p], grad_outputs=[go])
elif stage in ("st_param", "mt_param"):
    V, F = 500, 900
    zero = torch.randn(V, 3, dtype=torch.float64, device=dev); vec = torch.randn(V, 3, dtype=torch.float64, device=dev)
    faces = torch.randint(0, V, (F, 3), dtype=torch.int32, device=dev)
    q = torch.randn(V, dtype=torch.float64, device=dev, requires_grad=True)
    gf = torch.randn(F, 9, dtype=torch.float64, device=dev)
    def h():
        fv, nrm = ops.param_faces(q, zero, vec, faces)
        return fv
    for _ in range(3):
        with torch.autograd.set_multithreading_enabled(stage == "mt_param"):
            torch.autograd.grad([h()], [q], grad_outputs=[gf])
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        with torch.autograd.set_multithreading_enabled(stage == "mt_param"):
            gr = torch.autograd.grad([h()], [q], grad_outputs=[gf])
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed ok", float(gr[0].sum()))
