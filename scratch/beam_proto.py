"""CPU prototype: how many nodes / faces does the beam of a 64-ray coherent wave touch on the
bench scene (pass 0)?  python scratch/beam_proto.py [n_rays]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import scene_util
from tensorflowraytrace_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
sc = scene_util.lens_scene(N, k_front=41, k_back=9)
def tris(zero, faces, p):
    v = zero + p[:, None] * sc["vector"][None, :]
    return v[faces]
F = np.concatenate([tris(sc["zero_f"], sc["faces_f"], sc["p_f"]), tris(sc["zero_b"], sc["faces_b"], sc["p_b"]),
                    sc["target_verts"][sc["target_faces"]]])          # (M,3,3)
M = F.shape[0]
order = ops.cluster_order(torch.tensor(F.reshape(M, 9))).numpy()
def tri_sphere(T):
    A, B, C = T[:, 0], T[:, 1], T[:, 2]
    ab, ac, bc = B - A, C - A, C - B
    dA = (ab * ac).sum(1); dB = -(ab * bc).sum(1); dC = (ac * bc).sum(1)
    n = np.cross(ab, ac); n2 = (n * n).sum(1)
    off = ((ac * ac).sum(1)[:, None] * np.cross(n, ab) + (ab * ab).sum(1)[:, None] * np.cross(ac, n)) / (2 * n2[:, None])
    c = A + off; r = np.sqrt((off * off).sum(1))
    for cond, P, Q in ((dA <= 0, B, C), (dB <= 0, A, C), (dC <= 0, A, B)):
        c[cond] = 0.5 * (P + Q)[cond]; r[cond] = 0.5 * np.linalg.norm((P - Q)[cond], axis=1)
    return c, r
fc, fr = tri_sphere(F)
pad = (-M) % 16
mem = np.concatenate([order, -np.ones(pad, dtype=np.int64)])
ncl = mem.size // 16
def ball(points):     # Badoiu-Clarkson
    c = points.mean(0)
    for it in range(1, 30):
        d = np.linalg.norm(points - c, axis=1); c = c + (points[d.argmax()] - c) / (it + 1)
    return c, np.linalg.norm(points - c, axis=1).max()
clc = np.zeros((ncl, 3)); clr = np.zeros(ncl)
for k in range(ncl):
    ids = mem[16 * k:16 * k + 16]; ids = ids[ids >= 0]
    clc[k], clr[k] = ball(F[ids].reshape(-1, 3))
nsu = (ncl + 7) // 8
suc = np.zeros((nsu, 3)); sur = np.zeros(nsu)
for s in range(nsu):
    ids = mem[128 * s:128 * s + 128]; ids = ids[ids >= 0]
    suc[s], sur[s] = ball(F[ids].reshape(-1, 3))
rays = sc["rays"]; s0 = rays[:3].T; e0 = rays[3:].T
u = (e0 - s0); u /= np.linalg.norm(u, axis=1, keepdims=True)
# coherent order: Morton code of the aperture point (y, z at x = 0)
def morton2(a, b):
    def spread(v):
        v = v.astype(np.uint64); v = (v | (v << 16)) & 0x0000FFFF0000FFFF; v = (v | (v << 8)) & 0x00FF00FF00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F; v = (v | (v << 2)) & 0x3333333333333333; v = (v | (v << 1)) & 0x5555555555555555
        return v
    return spread(a) | (spread(b) << 1)
def hilbert2(x, y, bits=16):
    x = x.astype(np.int64).copy(); y = y.astype(np.int64).copy(); d = np.zeros_like(x)
    n = 1 << bits
    s = n >> 1
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        swap = ry == 0
        flip = swap & (rx == 1)
        x = np.where(flip, n - 1 - x, x); y = np.where(flip, n - 1 - y, y)
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d
q = lambda x: np.clip(((x + 1) * 0.5 * 65535), 0, 65535)
for name, perm in (("morton", np.argsort(morton2(q(e0[:, 1]), q(e0[:, 2])))),
                   ("hilbert", np.argsort(hilbert2(q(e0[:, 1]), q(e0[:, 2]))))):
    stats = []
    rng = np.random.default_rng(0)
    for w in rng.choice(N // 64, 600, replace=False):
        idx = perm[64 * w:64 * w + 64]
        S, U = s0[idx], u[idx]
        wdir = U.sum(0); wdir /= np.linalg.norm(wdir); o = S.mean(0)
        cosk = U @ wdir
        m = (U - cosk[:, None] * wdir) / cosk[:, None]
        d = S - o; ts = d @ wdir; A = d - ts[:, None] * wdir - ts[:, None] * m
        R0 = np.linalg.norm(A, axis=1).max(); Sl = np.linalg.norm(m, axis=1).max()
        def touched(c, r):
            v = c - o; t = v @ wdir; d2 = np.maximum((v * v).sum(1) - t * t, 0)
            B = r + R0 + Sl * (np.abs(t) + r)
            return d2 <= B * B
        ts_ = touched(suc, sur); sl = np.nonzero(ts_)[0]
        cl = np.concatenate([np.arange(8 * s, min(8 * s + 8, ncl)) for s in sl]) if sl.size else np.zeros(0, int)
        tc = cl[touched(clc[cl], clr[cl])] if cl.size else cl
        ms = np.concatenate([mem[16 * k:16 * k + 16] for k in tc]) if tc.size else np.zeros(0, int)
        ms = ms[ms >= 0]
        tf = ms[touched(fc[ms], fr[ms])] if ms.size else ms
        # per-ray sphere survivors among the candidate faces (what the screen would mostly reject)
        stats.append((sl.size, tc.size, tf.size, R0, Sl))
    st = np.array(stats)
    print(f"{name:8s} N={N} M={M}: touched supers {st[:,0].mean():.1f} (max {st[:,0].max():.0f}), clusters {st[:,1].mean():.1f} (max {st[:,1].max():.0f}), "
          f"candidate faces {st[:,2].mean():.1f} (max {st[:,2].max():.0f}); R0 {st[:,3].mean():.4f} S {st[:,4].mean():.5f}")
    print("   candidate-face percentiles 50/90/99:", np.percentile(st[:, 2], [50, 90, 99]), " clusters:", np.percentile(st[:, 1], [50, 90, 99]))
