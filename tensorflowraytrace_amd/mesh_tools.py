"""
Minimal mesh support for triangle boundaries (stand-in for the pyvista objects the reference
uses; pyvista/VTK are not dependencies of this package).

* ``PolyData``        points (V,3) float64 + faces in the VTK flat layout ``[3,i,j,k, 3,...]``;
                      ``copy``, ``rotate_x/y/z`` (degrees, like pyvista), ``save``/``read`` (ASCII STL)
* ``hexagonal_mesh``  same vertex order and triangulation as tfrt/mesh_tools.py:713-795
* ``plane``           the 2-triangle target used by dev/hexalens.py:99-106 (pv.Plane(...).triangulate())
* ``get_closest_point``  tfrt/mesh_tools.py:75-80
"""
import math

import numpy as np

PI = math.pi


class PolyData:
    def __init__(self, points=None, faces=None):
        if isinstance(points, PolyData):
            faces = points.faces.copy()
            points = points.points.copy()
        self.points = np.array(points, dtype=np.float64).reshape(-1, 3) if points is not None \
            else np.zeros((0, 3))
        self.faces = np.array(faces, dtype=np.int64).reshape(-1) if faces is not None \
            else np.zeros((0,), dtype=np.int64)

    @property
    def n_points(self):
        return self.points.shape[0]

    @property
    def n_faces(self):
        return self.faces.shape[0] // 4

    def triangles(self):
        f = self.faces.reshape(-1, 4)
        if f.size and not np.all(f[:, 0] == 3):
            raise ValueError("TriangleBoundary: mesh must consist entirely of triangles.")
        return f[:, 1:]

    def copy(self):
        return PolyData(self.points.copy(), self.faces.copy())

    def _rotate(self, axis, angle_deg):
        a = math.radians(angle_deg)
        c, s = math.cos(a), math.sin(a)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        p = self.points.copy()
        p[:, i] = c * self.points[:, i] - s * self.points[:, j]
        p[:, j] = s * self.points[:, i] + c * self.points[:, j]
        self.points = p
        return self

    def rotate_x(self, angle):
        return self._rotate(0, angle)

    def rotate_y(self, angle):
        return self._rotate(1, angle)

    def rotate_z(self, angle):
        return self._rotate(2, angle)

    def translate(self, offset):
        self.points = self.points + np.asarray(offset, dtype=np.float64).reshape(1, 3)
        return self

    def triangulate(self):
        return self

    def save(self, filename, **kwargs):
        """ASCII STL (what boundary.save wrote through pyvista, boundaries.py:872-874)."""
        tri = self.points[self.triangles()]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        n = np.divide(n, ln, out=np.zeros_like(n), where=ln > 0)
        with open(filename, "w") as f:
            f.write("solid tfrt\n")
            for nn, t in zip(n, tri):
                f.write(f"facet normal {nn[0]:.17g} {nn[1]:.17g} {nn[2]:.17g}\n outer loop\n")
                for v in t:
                    f.write(f"  vertex {v[0]:.17g} {v[1]:.17g} {v[2]:.17g}\n")
                f.write(" endloop\nendfacet\n")
            f.write("endsolid tfrt\n")


def read(filename):
    """Read an ASCII STL written by ``PolyData.save`` (vertices are merged exactly)."""
    verts, index, faces = [], {}, []
    cur = []
    with open(filename) as f:
        for line in f:
            parts = line.split()
            if parts[:1] == ["vertex"]:
                key = tuple(float(x) for x in parts[1:4])
                if key not in index:
                    index[key] = len(verts)
                    verts.append(key)
                cur.append(index[key])
                if len(cur) == 3:
                    faces.append([3] + cur)
                    cur = []
    return PolyData(np.array(verts).reshape(-1, 3), np.array(faces, dtype=np.int64).reshape(-1))


def get_closest_point(mesh, target):
    d = np.sum((mesh.points - np.asarray(target, dtype=np.float64)) ** 2, axis=1)
    return int(np.argmin(d))


def hexagonal_mesh(radius=1.0, step_count=10):
    """Hexagon of equilateral triangles in the x-y plane: 6*step_count^2 faces,
    3k^2+3k+1 vertices, ordered centre first then ring by ring, each ring walking its six
    edges counter-clockwise from angle 0 (the reference's vertex order)."""
    k = int(step_count)
    points = [(0.0, 0.0, 0.0)]
    ring_start = [0]
    for r in range(1, k + 1):
        rad = radius * r / k
        ring_start.append(len(points))
        for t in range(6):
            a0, a1 = PI / 3 * t, PI / 3 * (t + 1)
            p0 = np.array([rad * math.cos(a0), rad * math.sin(a0), 0.0])
            p1 = np.array([rad * math.cos(a1), rad * math.sin(a1), 0.0])
            for m in range(r):
                points.append(tuple(p0 + (p1 - p0) * (m / r)))

    def ring_index(r, pos):
        if r == 0:
            return 0
        return ring_start[r] + (pos % (6 * r))

    faces = []
    for r in range(1, k + 1):
        for t in range(6):
            for m in range(r):
                o0 = ring_index(r, t * r + m)
                o1 = ring_index(r, t * r + m + 1)
                i0 = ring_index(r - 1, t * (r - 1) + m)
                faces.append((3, o0, o1, i0))
                if m < r - 1:
                    i1 = ring_index(r - 1, t * (r - 1) + m + 1)
                    faces.append((3, i0, o1, i1))
    return PolyData(np.array(points), np.array(faces, dtype=np.int64).reshape(-1))


def plane(center=(0, 0, 0), direction=(1, 0, 0), i_size=1.0, j_size=1.0):
    """Two triangles spanning a rectangle centred at ``center`` with normal ``direction``."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    helper = np.array([0.0, 0.0, 1.0]) if abs(d[2]) < 0.9 else np.array([0.0, 1.0, 0.0])
    u = np.cross(helper, d)
    u /= np.linalg.norm(u)
    v = np.cross(d, u)
    c = np.asarray(center, dtype=np.float64)
    hu, hv = 0.5 * i_size * u, 0.5 * j_size * v
    pts = np.array([c - hu - hv, c + hu - hv, c + hu + hv, c - hu + hv])
    return PolyData(pts, np.array([3, 0, 1, 2, 3, 0, 2, 3], dtype=np.int64))
