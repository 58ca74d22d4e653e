"""
Deterministic synthetic scenes shared by the parity tests, the golden-vector generator and
bench.py.  Pure numpy; no oracle, no product imports.

* ``hex_mesh(k)``           6k^2 equilateral triangles / 3k^2+3k+1 vertices on a unit hexagon
                            (same counts as the reference's mesh_tools.hexagonal_mesh,
                            tfrt/mesh_tools.py:713-795; vertex order differs, which is
                            irrelevant to parity since both sides consume the same arrays)
* ``sunflower(n, radius)``  golden-spiral disc, tfrt/distributions.py:1574-1582
* ``lens_scene(...)``       two-surface parametric acrylic lens + square target + aperture
                            source: the cfg2/cfg3/cfg4 workload
"""
import math

import numpy as np

PI = math.pi


def hex_mesh(k, radius=1.0):
    """Returns points (V,2) in the mesh plane and faces (F,3) int32, CCW."""
    idx = {}
    pts = []
    for i in range(-k, k + 1):
        for j in range(-k, k + 1):
            if abs(i + j) <= k:
                idx[(i, j)] = len(pts)
                a = (i + 0.5 * j) * radius / k
                b = (math.sqrt(3.0) / 2.0 * j) * radius / k
                pts.append((a, b))
    faces = []
    for i in range(-k - 1, k + 1):
        for j in range(-k - 1, k + 1):
            up = ((i, j), (i + 1, j), (i, j + 1))
            down = ((i + 1, j), (i + 1, j + 1), (i, j + 1))
            for tri in (up, down):
                if all(t in idx for t in tri):
                    faces.append([idx[t] for t in tri])
    return np.asarray(pts, dtype=np.float64), np.asarray(faces, dtype=np.int32)


def sunflower(n, radius):
    i = np.arange(n, dtype=np.float64) + 0.5
    r = np.sqrt(i / n)
    theta = PI * (1 + 5 ** 0.5) * i
    return radius * np.stack([r * np.cos(theta), r * np.sin(theta)], axis=1)


def lens_scene(n_rays, k_front=9, k_back=9, wavelength=575.0, source_distance=10.0,
               target_distance=10.0, object_radius=0.2, aperture=0.98, sag=0.15, edge=0.1,
               seed=None):
    """Two hex-mesh surfaces displaced along x (vector generator (1,0,0)), acrylic inside.

    Returns a dict of numpy arrays:
      zero_f / zero_b (V,3) zero points, faces_f / faces_b (F,3) int32 (front faces are
      reversed so its norm points to -x, like flip_norm=True, boundaries.py:1096-1101),
      p_f / p_b (V,) parameters, vectors (1,0,0), target_verts (4,3), target_faces (2,3),
      rays (6,N) float64 [xs,ys,zs,xe,ye,ze], wavelength (N,), goal (N,2).
    """
    def surface(k, flip, sign):
        pts, faces = hex_mesh(k)
        zero = np.stack([np.zeros(len(pts)), pts[:, 0], pts[:, 1]], axis=1)
        r2 = pts[:, 0] ** 2 + pts[:, 1] ** 2
        p = sign * (edge + sag * (1.0 - r2))
        if flip:
            faces = faces[:, ::-1].copy()
        return zero, faces, p

    zero_f, faces_f, p_f = surface(k_front, True, -1.0)
    zero_b, faces_b, p_b = surface(k_back, False, +1.0)
    t = target_distance
    target_verts = np.array([[t, -50.0, -50.0], [t, 50.0, -50.0], [t, 50.0, 50.0], [t, -50.0, 50.0]])
    target_faces = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)

    if seed is None:
        start2 = sunflower(n_rays, object_radius)
        end2 = sunflower(n_rays, aperture)
    else:
        rng = np.random.default_rng(seed)
        def disc(r):
            rr = r * np.sqrt(rng.uniform(size=n_rays))
            th = 2 * PI * rng.uniform(size=n_rays)
            return np.stack([rr * np.cos(th), rr * np.sin(th)], axis=1)
        start2, end2 = disc(object_radius), disc(aperture)
    rays = np.stack([
        np.full(n_rays, -source_distance), start2[:, 0], start2[:, 1],
        np.zeros(n_rays), end2[:, 0], end2[:, 1]], axis=0)
    goal = -start2  # magnification 1 image of the object point (dev/hexalens.py:154-157)
    return dict(
        zero_f=zero_f, faces_f=faces_f, p_f=p_f, zero_b=zero_b, faces_b=faces_b, p_b=p_b,
        vector=np.array([1.0, 0.0, 0.0]), target_verts=target_verts, target_faces=target_faces,
        rays=rays, wavelength=np.full(n_rays, wavelength), goal=goal,
    )
