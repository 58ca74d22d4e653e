import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
from oracle import tracer
dev = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
n_faces = int(rng.choice([64, 97, 300, 640])); n_rays = int(rng.choice([50, 700, 2500]))
scale = 10 ** rng.uniform(-3, 3)
offset = rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(0, 2.5) * (rng.random() < 0.5)
centre = rng.uniform(-1, 1, (n_faces, 1, 3)); size = 10 ** rng.uniform(-2.5, -0.2, (n_faces, 1, 1))
tri = (centre + size * rng.standard_normal((n_faces, 3, 3))) * scale + offset
if rng.random() < 0.5: tri[: n_faces // 3, :, 2] = offset[2] + 0.1 * scale
P = torch.tensor(tri.reshape(n_faces, 9), dtype=torch.float64)
cat = torch.zeros(n_faces, dtype=torch.int64); cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1; cat[int(0.9 * n_faces):] = 2
n_in = torch.tensor(rng.uniform(1.0, 1.7, n_faces)); n_out = torch.tensor(rng.uniform(1.0, 1.7, n_faces))
s = rng.uniform(-1.5, 1.5, (3, n_rays)) * scale + offset[:, None]
d = rng.standard_normal((3, n_rays))
if rng.random() < 0.5: d[2] *= 1e-3
e = s + d * scale * 10 ** rng.uniform(-2, 0.5)
rays = torch.tensor(np.concatenate([s, e]), dtype=torch.float64)
fv = P.to(dev); L = float(scale)
def run(order):
    sc = ops.Scene3DArgs(fv, cat.int().to(dev), n_in=n_in.to(dev), n_out=n_out.to(dev), cluster_order=order)
    return ops.trace3d(rays.to(dev), fv, sc, max_passes=3, flags=flags, new_ray_length=L)
g = run(ops.cluster_order(fv)); a = run(None)
print("group == all-pairs:", all(torch.equal(g[c], a[c]) and torch.equal(g[c + "_face"], a[c + "_face"]) for c in ("finished", "active", "stopped", "dead")))
def sub(mask):
    verts = P[mask].reshape(-1, 3)
    dd = tracer.faces_from_vertices(verts, torch.arange(verts.shape[0]).reshape(-1, 3))
    dd["n_in"] = n_in[mask]; dd["n_out"] = n_out[mask]; return dd
system = tracer.System(3, optical=sub(cat == 0), stop=sub(cat == 1), target=sub(cat == 2))
src = {n: rays[i] for i, n in enumerate(("x_start", "y_start", "z_start", "x_end", "y_end", "z_end"))}
src["ray_id"] = torch.arange(n_rays, dtype=torch.float64)
ref = tracer.ray_trace(system, src, max_iterations=3, inherit=("ray_id",), index_type="value", new_ray_length=L,
                       flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
r = ref["dead"]
got = a["dead"].cpu().numpy(); want = np.stack([r[f].numpy() for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")])
err = np.abs(got - want).max(0)
bad = np.nonzero(err > 1e-9 * np.abs(want).max())[0]
print("dead rows differing:", bad, "ids", a["dead_id"].cpu().numpy()[bad])
for k in bad[:3]:
    rid = int(a["dead_id"][k])
    print("ray", rid, "src", rays[:, rid].numpy())
    print(" gpu   ", got[:, k]); print(" oracle", want[:, k])
    # history of that ray in the active set (both)
    for name, src_ in (("gpu", a), ("oracle", None)):
        if src_ is not None:
            m = (a["active_id"] == rid).cpu().numpy()
            print("  gpu active rows:", a["active"].cpu().numpy()[:, m].T, "faces", a["active_face"].cpu().numpy()[m])
        else:
            ra = ref["active"]; m = (ra["ray_id"].numpy() == rid)
            print("  oracle active rows:", np.stack([ra[f].numpy()[m] for f in ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")]).T)
# details of the second reaction of the first differing ray
k = bad[0]; rid = int(a["dead_id"][k])
m = (a["active_id"] == rid).cpu().numpy()
rows = a["active"].cpu().numpy()[:, m].T; fcs = a["active_face"].cpu().numpy()[m]
s2, h2, f2 = rows[-1][:3], rows[-1][3:], int(fcs[-1])
Pf = P[f2].numpy().reshape(3, 3)
C = np.cross(Pf[1] - Pf[0], Pf[2] - Pf[1]); N = C / np.linalg.norm(C)
u = (h2 - s2) / np.linalg.norm(h2 - s2); nu = float(N @ u)
ni, no = float(n_in[f2]), float(n_out[f2])
eta = ni / no if nu > 0 else no / ni
kk = 1 - eta * eta + (eta * nu) ** 2
print("face", f2, "cat", int(cat[f2]), "nu", nu, "n_in", ni, "n_out", no, "eta", eta, "radicand", kk)
refl = -2 * nu * N + u
refr = (np.sign(nu) * np.sqrt(max(kk, 0)) - eta * nu) * N + eta * u
print("reflect end", h2 + L * refl, "refract end", h2 + L * refr)
# which faces contain the hit point h2 (coplanar overlapping triangles tie in ray_u)?
def contains(Pf, h):
    a, b, c = Pf; n = np.cross(b - a, c - a); nn = n / np.linalg.norm(n)
    if abs((h - a) @ nn) > 1e-9 * max(1.0, np.abs(h).max()): return False
    def side(p, q): return np.cross(q - p, h - p) @ nn
    s1, s2_, s3 = side(a, b), side(b, c), side(c, a)
    return (s1 >= 0 and s2_ >= 0 and s3 >= 0) or (s1 <= 0 and s2_ <= 0 and s3 <= 0)
cands = [j for j in range(n_faces) if contains(P[j].numpy().reshape(3, 3), h2)]
print("faces containing the hit point:", cands)
want_end = want[3:, k]
for j in cands:
    Pf = P[j].numpy().reshape(3, 3); C = np.cross(Pf[1] - Pf[0], Pf[2] - Pf[1]); N = C / np.linalg.norm(C)
    nu = float(N @ u); ni, no = float(n_in[j]), float(n_out[j]); eta = ni / no if nu > 0 else no / ni
    kk = 1 - eta * eta + (eta * nu) ** 2
    w = (-2 * nu * N + u) if kk < 0 else ((np.sign(nu) * np.sqrt(kk) - eta * nu) * N + eta * u)
    print(" face", j, "cat", int(cat[j]), "radicand", kk, "end", h2 + L * w, "matches oracle:", np.allclose(h2 + L * w, want_end, atol=1e-6), "matches gpu:", np.allclose(h2 + L * w, got[3:, k], atol=1e-6))
# exact ray_u of those faces for the pass-2 input ray, by the kernel seam (float64, all pairs)
