"""CPU estimate of the filter funnel: cluster hits / member-sphere hits per ray (pass 1)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import scene_util
from tensorflowraytrace_amd import ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = scene_util.lens_scene(3000, k_front=41, k_back=9)
def faces(zero, fidx, p):
    v = zero + np.outer(p, sc["vector"])
    return v[fidx].reshape(-1, 9)
fv = np.concatenate([faces(sc["zero_f"], sc["faces_f"], sc["p_f"]), faces(sc["zero_b"], sc["faces_b"], sc["p_b"]),
                     sc["target_verts"][sc["target_faces"]].reshape(-1, 9)])
M = fv.shape[0]
order = ops.morton_order(torch.tensor(fv)).numpy()
A, B, C = fv[:, 0:3], fv[:, 3:6], fv[:, 6:9]
# bounding sphere per face: circumsphere for acute, longest-edge sphere otherwise (approx: use min enclosing via candidates)
def face_sphere(A, B, C):
    ab, ac, bc = B - A, C - A, C - B
    dA = (ab * ac).sum(1); dB = -(ab * bc).sum(1); dC = (ac * bc).sum(1)
    c = np.zeros_like(A); r2 = np.zeros(len(A))
    m = dA <= 0; c[m] = 0.5 * (B + C)[m]; r2[m] = 0.25 * (bc * bc).sum(1)[m]
    m2 = (~m) & (dB <= 0); c[m2] = 0.5 * (A + C)[m2]; r2[m2] = 0.25 * (ac * ac).sum(1)[m2]
    m3 = (~m) & (~m2) & (dC <= 0); c[m3] = 0.5 * (A + B)[m3]; r2[m3] = 0.25 * (ab * ab).sum(1)[m3]
    m4 = ~(m | m2 | m3)
    n = np.cross(ab, ac); n2 = (n * n).sum(1)
    t1 = np.cross(n, ab); t2 = np.cross(ac, n)
    off = ((ac * ac).sum(1)[:, None] * t1 + (ab * ab).sum(1)[:, None] * t2) / (2 * n2[:, None])
    c[m4] = (A + off)[m4]; r2[m4] = (off * off).sum(1)[m4]
    return c, np.sqrt(r2)
fc, fr = face_sphere(A, B, C)
ncl = (M + G - 1) // G
cc = np.zeros((ncl, 3)); cr = np.zeros(ncl)
for k in range(ncl):
    idx = order[k * G:(k + 1) * G]
    mean = fc[idx].mean(0)
    cc[k] = mean; cr[k] = (np.linalg.norm(fc[idx] - mean, axis=1) + fr[idx]).max()
rays = sc["rays"]  # (6,N)
s, e = rays[:3].T, rays[3:].T
u = (e - s); u /= np.linalg.norm(u, axis=1, keepdims=True)
def line_dist(c):  # (K,3) -> (N,K)
    w = c[None] - s[:, None]
    t = (w * u[:, None]).sum(2)
    return np.sqrt(np.maximum((w * w).sum(2) - t * t, 0))
dcl = line_dist(cc) <= cr[None]
print("G", G, "clusters", ncl, "median cluster radius", np.median(cr), "median face radius", np.median(fr))
print("cluster hits per ray: mean", dcl.sum(1).mean(), "max", dcl.sum(1).max())
dfa = line_dist(fc) <= fr[None]
print("face-sphere hits per ray: mean", dfa.sum(1).mean())
# member hits restricted to hit clusters
inv = np.empty(M, int); inv[order] = np.arange(M)
cl_of_face = inv // G
both = dfa & dcl[:, cl_of_face]
print("member hits via clusters:", both.sum(1).mean(), "(must equal face-sphere hits)")
print("cluster radius percentiles", np.percentile(cr, [10, 50, 90, 99, 100]).round(3), "sum r^2", (cr**2).sum())
hits_by_cluster = dcl.sum(0)
big = np.argsort(-cr)[:8]
print("largest clusters r:", cr[big].round(2), "their hit rates", (hits_by_cluster[big] / dcl.shape[0]).round(2))
print("hits from clusters with r>0.2:", dcl[:, cr > 0.2].sum(1).mean(), " n such", (cr > 0.2).sum())
