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

    def save(self, filename, binary=True, **kwargs):
        """Write an STL file: binary little-endian by default (what ``boundary.save`` produced
        through pyvista, boundaries.py:872-874), ASCII with ``binary=False``."""
        tri = self.points[self.triangles()]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        n = np.divide(n, ln, out=np.zeros_like(n), where=ln > 0)
        if binary:
            rec = np.zeros(tri.shape[0], dtype=_STL_RECORD)
            rec["normal"] = n
            rec["v"] = tri
            with open(filename, "wb") as f:
                f.write(b"tfrt binary STL".ljust(80, b" "))
                f.write(np.uint32(tri.shape[0]).tobytes())
                f.write(rec.tobytes())
            return
        with open(filename, "w") as f:
            f.write("solid tfrt\n")
            for nn, t in zip(n, tri):
                f.write(f"facet normal {nn[0]:.17g} {nn[1]:.17g} {nn[2]:.17g}\n outer loop\n")
                for v in t:
                    f.write(f"  vertex {v[0]:.17g} {v[1]:.17g} {v[2]:.17g}\n")
                f.write(" endloop\nendfacet\n")
            f.write("endsolid tfrt\n")


# binary STL: 80-byte header, uint32 count, then 50-byte records (float32 normal, 3 float32
# vertices, uint16 attribute)
_STL_RECORD = np.dtype([("normal", "<f4", (3,)), ("v", "<f4", (3, 3)), ("attr", "<u2")])


def _merge_vertices(tri):
    """(F,3,3) corner coordinates -> PolyData with exactly equal corners merged, first
    occurrence order (so a mesh written by ``PolyData.save`` reads back with its topology)."""
    flat = tri.reshape(-1, 3)
    uniq, first, inverse = np.unique(flat, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)                    # np.unique sorts by value: undo that
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    idx = rank[inverse.reshape(-1)].reshape(-1, 3)
    faces = np.concatenate([np.full((idx.shape[0], 1), 3, dtype=np.int64), idx], axis=1)
    return PolyData(uniq[order].astype(np.float64), faces.reshape(-1))


def read(filename):
    """Read an STL file, binary or ASCII (the reference reads meshes through ``pv.read``,
    boundaries.py:859-861).  Binary is recognised by its size: 84 + 50 * n_facets bytes."""
    with open(filename, "rb") as f:
        blob = f.read()
    if len(blob) >= 84:
        count = int(np.frombuffer(blob[80:84], dtype="<u4")[0])
        if len(blob) == 84 + 50 * count:
            rec = np.frombuffer(blob[84:], dtype=_STL_RECORD, count=count)
            return _merge_vertices(rec["v"].astype(np.float64))
    corners = []
    for line in blob.decode("ascii", errors="replace").splitlines():
        parts = line.split()
        if parts[:1] == ["vertex"]:
            corners.append([float(x) for x in parts[1:4]])
    if len(corners) % 3:
        raise ValueError(f"{filename}: ASCII STL with a vertex count that is not a multiple of 3")
    return _merge_vertices(np.array(corners, dtype=np.float64).reshape(-1, 3, 3))


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


def sphere(radius=0.5, center=(0, 0, 0), theta_resolution=30, phi_resolution=30):
    """Latitude/longitude triangulation of a sphere, poles on the z axis (the role of
    ``pv.Sphere`` in dev/3d_trace.py:32): ``theta_resolution`` meridians,
    ``phi_resolution`` latitude rings including the two poles.  Outward-facing triangles."""
    nt, nphi = int(theta_resolution), int(phi_resolution)
    if nt < 3 or nphi < 3:
        raise ValueError("sphere: theta_resolution and phi_resolution must be >= 3")
    theta = np.linspace(0.0, 2 * PI, nt, endpoint=False)
    phi = np.linspace(0.0, PI, nphi)[1:-1]
    ring = np.stack([np.outer(np.sin(phi), np.cos(theta)), np.outer(np.sin(phi), np.sin(theta)),
                     np.outer(np.cos(phi), np.ones(nt))], axis=2).reshape(-1, 3)
    pts = np.concatenate([[[0.0, 0.0, 1.0]], ring, [[0.0, 0.0, -1.0]]]) * radius
    pts = pts + np.asarray(center, dtype=np.float64)
    south = pts.shape[0] - 1
    at = lambda r, t: 1 + r * nt + (t % nt)
    faces = []
    for t in range(nt):
        faces.append((0, at(0, t), at(0, t + 1)))
        for r in range(nphi - 3):
            faces.append((at(r, t), at(r + 1, t), at(r + 1, t + 1)))
            faces.append((at(r, t), at(r + 1, t + 1), at(r, t + 1)))
        faces.append((south, at(nphi - 3, t + 1), at(nphi - 3, t)))
    faces = np.array(faces, dtype=np.int64)
    cells = np.concatenate([np.full((faces.shape[0], 1), 3, dtype=np.int64), faces], axis=1)
    return PolyData(pts, cells.reshape(-1))


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


# ------------------------------------------------------------------------------------------
# more generators (tfrt/mesh_tools.py:576-952)

def circular_mesh(radius, target_edge_size, starting_radius=0, theta_start=0, theta_end=2 * PI,
                  join=None):
    """Disc / annulus / wedge of near-equilateral triangles in the x-y plane
    (tfrt/mesh_tools.py:576-711): rings at radii linspace(starting_radius, radius, n), every
    ring has ``trapezoid_count`` more points than the previous one; same vertex order as the
    reference."""
    if join is None:
        join = bool(theta_start == 0) and bool(theta_end == 2 * PI)
    if starting_radius >= radius:
        raise ValueError("circular_mesh: starting_radius must be < radius.")
    radius_step = target_edge_size * math.sin(PI / 3)
    n_rings = max(int(1 + (radius - starting_radius) / radius_step), 2)
    radii = np.linspace(starting_radius, radius, n_rings)
    traps = math.ceil((theta_end - theta_start) / (PI / 3))
    if starting_radius != 0:
        arc = radii[0] * (theta_end - theta_start) / traps
        inner_edge_points = math.ceil(arc / target_edge_size) + 1
    else:
        inner_edge_points = 1
    linear_count = (inner_edge_points - 1) * traps + 1  # points along a ring incl. both ends
    angles = np.linspace(theta_start, theta_end, linear_count)
    points = [(radii[0] * math.cos(a), radii[0] * math.sin(a), 0.0) for a in angles]
    rings = [(0, len(points), linear_count)]  # (first index, stored count, linear count)
    faces = []
    for r in radii[1:]:
        linear_count += traps
        ang = np.linspace(theta_start, theta_end, linear_count)
        new = [(r * math.cos(a), r * math.sin(a), 0.0) for a in ang]
        if join:
            new.pop()
        first = len(points)
        points += new
        rings.append((first, len(new), linear_count))
        (i0, ic, il), (o0, oc, ol) = rings[-2], rings[-1]
        e_in = (il - 1) // traps      # edges per trapezoid on the inner ring
        e_out = (ol - 1) // traps
        assert e_out == e_in + 1
        inner = lambda k: i0 + (k % ic if ic > 0 else 0)
        outer = lambda k: o0 + (k % oc)
        for t in range(traps):
            for m in range(e_out):
                o_a, o_b = outer(t * e_out + m), outer(t * e_out + m + 1)
                i_a = inner(t * e_in + min(m, e_in))
                faces.append((3, o_a, o_b, i_a))
                if m < e_in:
                    faces.append((3, i_a, o_b, inner(t * e_in + m + 1)))
    return PolyData(np.array(points), np.array(faces, dtype=np.int64).reshape(-1))


def cylindrical_mesh(start, end, radius=1.0, theta_res=6, z_res=8, start_cap=True, end_cap=True,
                     use_twist=False, epsilion=1e-6):
    """Closed cylinder between two axis points (tfrt/mesh_tools.py:800-952): optional cap
    centre vertices first/last, ``z_res`` rings of ``theta_res`` vertices in between."""
    start = np.reshape(np.asarray(start, dtype=np.float64), (1, 3))
    end = np.reshape(np.asarray(end, dtype=np.float64), (1, 3))
    axis = end - start
    u = np.cross(axis, (1.0, 0.0, 0.0))
    if np.linalg.norm(u) < epsilion:
        u = np.cross(axis, (0.0, 1.0, 0.0))
    if np.linalg.norm(u) < epsilion:
        raise ValueError("cylindrical_mesh: could not find vectors perpendicular to axis.  Try "
                         "decreasing epsilion?")
    u = (u * radius / np.linalg.norm(u)).reshape(1, 3)
    v = np.cross(axis, u)
    v = (v * radius / np.linalg.norm(v)).reshape(1, 3)
    theta, z = np.meshgrid(np.linspace(0, 2 * PI, theta_res + 1)[:-1], np.linspace(0, 1, z_res))
    if use_twist:
        theta = theta + np.reshape(PI / theta_res * np.arange(z_res), (-1, 1))
    ring_pts = (start + z[..., None] * axis + np.cos(theta)[..., None] * u
                + np.sin(theta)[..., None] * v).reshape(-1, 3)
    off = 1 if start_cap else 0
    points = ([start[0]] if start_cap else []) + list(ring_pts) + ([end[0]] if end_cap else [])
    th = np.arange(theta_res)
    nxt = (th + 1) % theta_res
    faces = []
    if start_cap:
        faces += [(t + 1, 0, n + 1) for t, n in zip(th, nxt)]
    for zz in range(1, z_res):
        lo, hi = (zz - 1) * theta_res + off, zz * theta_res + off
        for t, n in zip(th, nxt):
            faces.append((lo + n, hi + t, lo + t))
            faces.append((hi + t, lo + n, hi + n))
    if end_cap:
        last = len(points) - 1
        zo = (z_res - 1) * theta_res + off
        faces += [(n + zo, last, t + zo) for t, n in zip(th, nxt)]
    f = np.array(faces, dtype=np.int64)
    f = np.concatenate([np.full((f.shape[0], 1), 3, dtype=np.int64), f], axis=1)
    return PolyData(np.array(points), f.reshape(-1))


# ------------------------------------------------------------------------------------------
# parametrisation helpers (tfrt/mesh_tools.py:210-520)

def get_faces_as_sets(mesh):
    return [set(int(v) for v in face) for face in mesh.triangles()]


def _vertex_faces(mesh):
    vf = [[] for _ in range(mesh.n_points)]
    for fi, face in enumerate(mesh.triangles()):
        for v in face:
            vf[int(v)].append(fi)
    return vf


def _vertex_neighbors(mesh):
    nb = [set() for _ in range(mesh.n_points)]
    for face in mesh.triangles():
        a, b, c = (int(v) for v in face)
        nb[a] |= {b, c}
        nb[b] |= {a, c}
        nb[c] |= {a, b}
    return nb


def raw_mesh_parametrization_tools(mesh, top_parent):
    """Breadth-first sweep outward from ``top_parent`` (tfrt/mesh_tools.py:221-285): every face
    may move only the vertices that are farther out than the sweep front that first reached
    it; every vertex's ancestors are the front vertices that led to it."""
    face_sets = get_faces_as_sets(mesh)
    vfaces = _vertex_faces(mesh)
    neighbors = _vertex_neighbors(mesh)
    n_faces, n_points = len(face_sets), mesh.n_points
    face_movable = [set() for _ in range(n_faces)]
    to_visit = set(range(n_faces))
    active, last = {int(top_parent)}, set()
    available = set(range(n_points))
    parents = [set() for _ in range(n_points)]
    ancestors = [set() for _ in range(n_points)]
    missed = set(range(n_points))
    while to_visit and active:
        nxt, visited = set(), set()
        available -= active
        for v in active:
            for f in vfaces[v]:
                if f in to_visit:
                    movable = face_sets[f] & available
                    nxt |= movable
                    face_movable[f] = movable
                    visited.add(f)
        for v in active:
            missed.discard(v)
            parents[v] = neighbors[v] & last
            anc = set(parents[v])
            for p in parents[v]:
                anc |= ancestors[p]
            ancestors[v] = anc
        to_visit -= visited
        last, active = active, nxt
    for v in active:  # the outermost front never became "active" inside the loop
        if v in missed:
            missed.discard(v)
            parents[v] = neighbors[v] & last
            anc = set(parents[v])
            for p in parents[v]:
                anc |= ancestors[p]
            ancestors[v] = anc
    for v in list(missed):
        parents[v] = neighbors[v] - missed
        anc = set(parents[v])
        for p in parents[v]:
            anc |= ancestors[p]
        ancestors[v] = anc
    return face_movable, ancestors, parents, missed


def movable_to_updatable(mesh, face_movable_vertices):
    """(F,3) bool: which corners each face may move (tfrt/mesh_tools.py:459-486); a face that
    could move nothing may move everything."""
    faces = mesh.triangles()
    out = np.zeros(faces.shape, dtype=bool)
    orphaned = 0
    for f in range(faces.shape[0]):
        row = [int(v) in face_movable_vertices[f] for v in faces[f]]
        if not any(row):
            orphaned += 1
            row = [True] * 3
        out[f] = row
    if orphaned:
        print("Mesh parametrization tools: warning, found orphaned faces in mesh.")
    return out


def connections_to_array(connection_list, dtype=np.float64, inverse=True):
    """tfrt/mesh_tools.py:490-506: identity + 1 at (i, j) for every j connected to i."""
    size = len(connection_list)
    arr = np.eye(size, dtype=dtype)
    for i, row in enumerate(connection_list):
        if row:
            arr[i, list(row)] += 1
    return arr if inverse else arr.T


def mesh_parametrization_tools(mesh, top_parent, active_vertices=None):
    """Returns (vertex_update_map (F,3) bool, gradient accumulator (V,V) f64)
    (tfrt/mesh_tools.py:289-331)."""
    face_movable, ancestors, _parents, _missed = raw_mesh_parametrization_tools(mesh, top_parent)
    vertex_update_map = movable_to_updatable(mesh, face_movable)
    accumulator = connections_to_array(ancestors)
    if active_vertices is not None:
        kept = [i for i in range(accumulator.shape[0]) if i in set(active_vertices)]
        accumulator = accumulator[:, kept][kept, :]
    return vertex_update_map, accumulator


def mesh_smoothing_tool(mesh, weights, active_vertices=None):
    """(V,V) smoothing matrix (tfrt/mesh_tools.py:345-421): row i spreads weight[k] (normalised)
    evenly over the k-th ring of neighbours of vertex i."""
    neighbors = _vertex_neighbors(mesh)
    n = mesh.n_points
    weights = np.asarray(weights, dtype=np.float64)
    weights = weights / np.sum(weights)
    smoother = np.zeros((n, n), dtype=np.float64)
    for p in range(n):
        ring = {p}
        taken = {p}
        for order in range(len(weights)):
            if order > 0:
                new = set()
                for q in ring:
                    new |= neighbors[q]
                new -= taken
                ring = new
                taken |= new
            if ring:
                smoother[p, list(ring)] = weights[order] / len(ring)
    if active_vertices is not None:
        kept = [i for i in range(n) if i in set(active_vertices)]
        smoother = smoother[:, kept][kept, :]
    return smoother


def get_flat_initial(mesh, axis=0):
    """Zero one coordinate of the mesh and return it (tfrt/mesh_tools.py:423-455)."""
    if axis not in {0, 1, 2}:
        raise ValueError("get_flat_initial: axis must be in {0, 1, 2}.")
    init = mesh.points[:, axis].copy()
    mesh.points[:, axis] = 0.0
    return init
