import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import faulthandler; faulthandler.enable()
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
stage = sys.argv[1]; dev = "cuda:0"
V, F = 500, 900
zero = torch.randn(V, 3, dtype=torch.float64, device=dev); vec = torch.randn(V, 3, dtype=torch.float64, device=dev)
faces = torch.randint(0, V, (F, 3), dtype=torch.int32, device=dev)
q = torch.randn(V, dtype=torch.float64, device=dev, requires_grad=True)
gf = torch.randn(F, 9, dtype=torch.float64, device=dev)
keep = {}
def h():
    fv, nrm = ops.param_faces(q, zero, vec, faces)
    if "keep" in stage and not (capturing and "nostore" in stage):
        keep["fv"], keep["nrm"] = fv, nrm
    return fv
def run():
    fv = h()
    go = torch.ones_like(fv) if "ones" in stage else gf
    with torch.autograd.set_multithreading_enabled(False):
        return torch.autograd.grad([fv], [q], grad_outputs=[go], allow_unused="unused" in stage)
capturing = False
side = torch.cuda.Stream()
if "side" in stage:
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): run()
    torch.cuda.current_stream().wait_stream(side)
else:
    for _ in range(3): run()
torch.cuda.synchronize()
if "clear" in stage: keep.clear()
capturing = True
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side if "side" in stage else None):
    gr = run()
if "drop" in stage: keep.clear()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed ok", float(gr[0].sum()))
