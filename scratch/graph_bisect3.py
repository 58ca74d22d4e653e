import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import faulthandler; faulthandler.enable()
import numpy as np, torch
import tfrt.boundaries as boundaries, tfrt.engine as engine, tfrt.mesh_tools as mt
stage = sys.argv[1]; dev = "cuda:0"
def hexm(k):
    zp = mt.hexagonal_mesh(1.0, k); zp.rotate_y(90); zp.rotate_x(90); return zp
zp = hexm(3)
vmap = (np.random.default_rng(0).uniform(size=(zp.n_faces, 3)) > 0.25) if "vmap" in stage else None
if "multi" in stage:
    cons = [boundaries.ThicknessConstraint(0.0, "min"), boundaries.ThicknessConstraint(0.2, "min")] if "cons" in stage else [boundaries.NoConstraint(), boundaries.NoConstraint()]
    lens = boundaries.ParametricMultiTriangleBoundary(zp, boundaries.FromVectorVG((1, 0, 0)), cons, [True, False],
        initial_parameters=[-0.1, 0.1], material_list=[{"mat_in": 1, "mat_out": 0}] * 2, vertex_update_map=vmap)
    surfaces = lens.surfaces
else:
    surfaces = [boundaries.ParametricTriangleBoundary(zp, boundaries.FromVectorVG((1, 0, 0)), flip_norm=f, initial_parameters=s,
                material_dict={"mat_in": 1, "mat_out": 0}, vertex_update_map=vmap) for f, s in ((True, -0.1), (False, 0.1))]
system = engine.OpticalSystem3D()
system.optical = surfaces if "one" not in stage else surfaces[:1]
if "target" in stage:
    t = boundaries.ManualTriangleBoundary(mesh=mt.plane(center=(10, 0, 0), direction=(1, 0, 0), i_size=100, j_size=100)); t.frozen = True
    system.targets = [t]
params = [s.parameters for s in (surfaces if "one" not in stage else surfaces[:1])]
def run():
    system.update()
    fv = system._merged_face_verts
    go = torch.ones_like(fv)
    with torch.autograd.set_multithreading_enabled(False):
        return torch.autograd.grad([fv], params, grad_outputs=[go], allow_unused=True)
for _ in range(3): run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gr = run()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed ok", float(gr[0].sum()))
