"""CPU estimate of the hierarchy funnel with the shipped cluster order: supercluster / cluster /
member-sphere hits per ray for the three passes' typical rays (pass-1 rays only)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import scene_util
from tensorflowraytrace_amd import ops
sc = scene_util.lens_scene(3000, k_front=41, k_back=9)
def faces(zero, fidx, p):
    v = zero + np.outer(p, sc["vector"]); return v[fidx].reshape(-1, 9)
fv = np.concatenate([faces(sc["zero_f"], sc["faces_f"], sc["p_f"]), faces(sc["zero_b"], sc["faces_b"], sc["p_b"]),
                     sc["target_verts"][sc["target_faces"]].reshape(-1, 9)])
M = fv.shape[0]
order = ops.cluster_order(torch.tensor(fv)).numpy()
V = fv.reshape(M, 3, 3)
def meb(points, iters=12):
    c = points.mean(0)
    for it in range(1, iters + 1):
        d = np.linalg.norm(points - c, axis=1); c = c + (points[d.argmax()] - c) / (it + 1)
    return c, np.linalg.norm(points - c, axis=1).max()
ncl = (M + 15) // 16
cc = np.zeros((ncl, 3)); cr = np.zeros(ncl)
for k in range(ncl):
    idx = order[k * 16:(k + 1) * 16]
    cc[k], cr[k] = meb(V[idx].reshape(-1, 3))
ns = (ncl + 7) // 8
scn = np.zeros((ns, 3)); sr = np.zeros(ns)
for k in range(ns):
    sl = slice(k * 8, min(ncl, (k + 1) * 8))
    m = cc[sl].mean(0); scn[k] = m; sr[k] = (np.linalg.norm(cc[sl] - m, axis=1) + cr[sl]).max()
rays = sc["rays"]; s, e = rays[:3].T, rays[3:].T
u = e - s; u /= np.linalg.norm(u, axis=1, keepdims=True)
def hits(c, r):
    w = c[None] - s[:, None]; t = (w * u[:, None]).sum(2)
    return np.sqrt(np.maximum((w * w).sum(2) - t * t, 0)) <= r[None]
hs = hits(scn, sr); hc = hits(cc, cr)
print("superclusters", ns, "hit per ray", hs.sum(1).mean(), "| clusters", ncl, "hit per ray", hc.sum(1).mean())
print("largest supercluster radii", np.sort(sr)[-6:].round(2), "largest cluster radii", np.sort(cr)[-6:].round(2))
# level-1 rounds per tile: max over 64-ray waves of touched supers per 32-super tile
w = hs[: (hs.shape[0] // 64) * 64].reshape(-1, 64, ns)
rounds = sum(w[:, :, t:t + 32].sum(2).max(1).mean() for t in range(0, ns, 32))
print("level-1 rounds per wave (sum over tiles of max touched per lane)", rounds)
