"""Expected clusters touched per (axis-parallel) ray = sum of the clusters' bounding-disc areas
/ lens aperture area, for a given face order.  Compares the shipped k-d order with variants."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import scene_util
from tensorflowraytrace_amd import ops
sc = scene_util.lens_scene(100, k_front=41, k_back=9)
def faces(zero, fidx, p):
    v = zero + np.outer(p, sc["vector"]); return v[fidx].reshape(-1, 9)
fv = np.concatenate([faces(sc["zero_f"], sc["faces_f"], sc["p_f"]), faces(sc["zero_b"], sc["faces_b"], sc["p_b"]),
                     sc["target_verts"][sc["target_faces"]].reshape(-1, 9)])
M = fv.shape[0]
V = fv.reshape(M, 3, 3)
def meb(points, iters=60):
    c = points.mean(0)
    for it in range(1, iters + 1):
        d = np.linalg.norm(points - c, axis=1); c = c + (points[d.argmax()] - c) / (it + 1)
    return c, np.linalg.norm(points - c, axis=1).max()
def quality(order, G):
    n = (M + G - 1) // G
    area = 0.0; rs = []
    for k in range(n):
        idx = order[k * G:(k + 1) * G]
        idx = idx[idx < M]
        if len(idx) == 0: continue
        pts = V[idx].reshape(-1, 3)
        if np.abs(pts[:, 0]).max() > 5: continue        # target plane
        # rays run along x: project on (y, z)
        c, r = meb(pts[:, 1:])
        area += np.pi * r * r; rs.append(r)
    return area / np.pi, np.median(rs)
order = ops.cluster_order(torch.tensor(fv)).numpy()
for G in (16, 128):
    q, r = quality(order, G); print(f"shipped k-d order, groups of {G}: touched per ray (front+back) {q:.2f}, median radius {r:.4f}")

def balanced_kmeans(pts, k, size, iters=12, seed=0):
    n = len(pts)
    rng = np.random.default_rng(seed)
    # farthest-point init
    cent = [pts[rng.integers(n)]]
    for _ in range(k - 1):
        d = np.min([np.linalg.norm(pts - c, axis=1) for c in cent], axis=0)
        cent.append(pts[d.argmax()])
    cent = np.array(cent)
    assign = np.full(n, -1)
    for it in range(iters):
        d = np.linalg.norm(pts[:, None] - cent[None], axis=2)      # (n,k)
        # assign points in order of how much they "care" (gap between best and second best)
        assign[:] = -1; room = np.full(k, size)
        srt = np.sort(d, axis=1); pri = np.argsort(-(srt[:, 1] - srt[:, 0]))
        for p in pri:
            for c in np.argsort(d[p]):
                if room[c] > 0:
                    assign[p] = c; room[c] -= 1; break
        for c in range(k):
            if (assign == c).any(): cent[c] = pts[assign == c].mean(0)
    return assign

cent = V.mean(1)                                            # face centroids
new_order = order.copy()
G, S = 16, 128
for s0 in range(0, M, S):
    idx = order[s0:s0 + S]
    if len(idx) < S: continue
    if np.abs(V[idx][:, :, 0]).max() > 5: continue
    a = balanced_kmeans(cent[idx], S // G, G)
    new_order[s0:s0 + S] = np.concatenate([idx[a == c] for c in range(S // G)])
for Gq in (16, 128):
    q, r = quality(new_order, Gq); print(f"k-d supers + balanced k-means leaves, groups of {Gq}: touched per ray {q:.2f}, median radius {r:.4f}")
