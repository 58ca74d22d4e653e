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
                faces.append((3, o1, i0, o0))      # (the reference's corner order: third, second, first)
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
                faces.append((3, o_b, i_a, o_a))      # (the reference's corner order)
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


# ------------------------------------------------------------------------------------------
# small topology helpers (tfrt/mesh_tools.py:28-217, 335, 956-1160)

def pack_faces(faces):
    """(F,3) vertex indices -> the flat VTK layout ``[3,i,j,k, 3,...]`` (tfrt/mesh_tools.py:1143-1150)."""
    faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    return np.concatenate([np.full((faces.shape[0], 1), 3, dtype=np.int64), faces],
                          axis=1).reshape(-1)


def unpack_faces(faces):
    """Flat VTK layout -> (F,3) vertex indices (tfrt/mesh_tools.py:1152-1160)."""
    return np.asarray(faces, dtype=np.int64).reshape(-1, 4)[:, 1:]


def gaussian_weights(sigma, count):
    """Ring weights for ``mesh_smoothing_tool``: a half Gaussian sampled at 0..count-1
    (tfrt/mesh_tools.py:335-343); not normalised, the smoothing tool does that."""
    return np.exp(-0.5 * (np.arange(count, dtype=np.float64) / sigma) ** 2)


def get_unique_edges_1p(mesh):
    """Set of 2-element frozensets, one per undirected edge (tfrt/mesh_tools.py:84-102)."""
    tri = mesh.triangles()
    edges = set()
    for a, b in ((0, 1), (1, 2), (2, 0)):
        edges.update(frozenset((int(p), int(q))) for p, q in zip(tri[:, a], tri[:, b]))
    return edges


get_unique_edges = get_unique_edges_1p


def neighbors_from_edges(start, edges):
    """Vertices one edge away from ``start`` (tfrt/mesh_tools.py:105-115)."""
    out = set()
    for edge in edges:
        if start in edge:
            out |= edge
    out.discard(start)
    return out


def neighbors_from_faces(point, faces):
    """Same, from a list of face sets (tfrt/mesh_tools.py:119-130)."""
    out = set()
    for face in faces:
        if point in face:
            out |= set(face)
    out.discard(point)
    return out


def find_generations(top_parent, mesh):
    """Breadth-first rings around ``top_parent``: ``generations[k]`` is the set of vertices
    ``k`` edges away (tfrt/mesh_tools.py:195-217).  The sweep ends when a ring comes back empty,
    so vertices of other connected components are simply absent."""
    neighbors = _vertex_neighbors(mesh)
    seen = {int(top_parent)}
    generations = [set(seen)]
    while len(seen) < mesh.n_points:
        ring = set()
        for v in generations[-1]:
            ring |= neighbors[v]
        ring -= seen
        if not ring:
            break
        seen |= ring
        generations.append(ring)
    return generations


def find_all_relationships_1p(top_parent, mesh, edges):
    """Single-parent spanning tree of the mesh rooted at ``top_parent``
    (tfrt/mesh_tools.py:133-187): a breadth-first sweep in which the pending vertices are
    always expanded nearest-to-the-root first and every vertex is claimed by the first vertex
    that reaches it.  Returns per-vertex lists of sets (descendants, children, parents,
    ancestors)."""
    points = mesh.points
    n = points.shape[0]
    root = points[top_parent]
    adjacency = [set() for _ in range(n)]
    for edge in edges:
        a, b = tuple(edge)
        adjacency[a].add(b)
        adjacency[b].add(a)
    children = [set() for _ in range(n)]
    parents = [set() for _ in range(n)]
    ancestors = [set() for _ in range(n)]
    unclaimed = set(range(n)) - {top_parent}
    queue, order = [top_parent], []
    while queue:
        parent = queue.pop(0)
        order.append(parent)
        mine = adjacency[parent] & unclaimed
        unclaimed -= mine
        children[parent] = mine
        for child in mine:
            queue.append(child)
            parents[child] = {parent}
            ancestors[child] = {parent} | ancestors[parent]
        if queue:
            d = np.sum((points[queue] - root) ** 2, axis=1)
            queue = [queue[k] for k in np.argsort(d, kind="stable")]
    descendants = [set(c) for c in children]
    for v in reversed(order):
        for child in children[v]:
            descendants[v] |= descendants[child]
    return descendants, children, parents, ancestors


def gradient_accumulator_1p(mesh, origin=(0, 0, 0)):
    """Accumulator matrix from the single-parent tree around the vertex nearest ``origin``
    (tfrt/mesh_tools.py:28-72): row i sums the gradient of vertex i and all its descendants.
    Returns (accumulator (V,V), relationship dict)."""
    top_parent = get_closest_point(mesh, np.array(origin, dtype=np.float64))
    edges = get_unique_edges_1p(mesh)
    out = {"top_parent": top_parent, "unique_edges": edges}
    out["descendant"], out["child"], out["parent"], out["ancestor"] = \
        find_all_relationships_1p(top_parent, mesh, edges)
    return connections_to_array(out["descendant"]), out


def clean_mesh_raw(vertices, faces, distance_tolerance=1e-6):
    """Merge vertices whose *squared* distance is below ``distance_tolerance`` (that is the
    quantity the reference compares; the lowest index of a cluster survives), drop the merged
    vertices, then drop faces that became degenerate or that repeat an earlier face's vertex
    set; the surviving faces keep their winding (tfrt/mesh_tools.py:1073-1140).  The reference
    builds the dense (V,V) distance matrix; a k-d tree finds the same pairs in O(V log V).
    ``faces`` is (F,3).  Returns (vertices, faces)."""
    from scipy.spatial import cKDTree

    points = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    n = points.shape[0]
    target = np.arange(n)
    if n > 1:
        pairs = cKDTree(points).query_pairs(math.sqrt(distance_tolerance), output_type="ndarray")
        if pairs.size:      # the tree's test is <=, the reference's is <
            d2 = np.sum((points[pairs[:, 0]] - points[pairs[:, 1]]) ** 2, axis=1)
            pairs = pairs[d2 < distance_tolerance]
        # a vertex maps to the smallest index it is (transitively) within tolerance of
        if pairs.size:
            lo, hi = pairs.min(axis=1), pairs.max(axis=1)
            changed = True
            while changed:
                before = target.copy()
                np.minimum.at(target, hi, target[lo])
                np.minimum.at(target, lo, target[hi])
                target = target[target]
                changed = not np.array_equal(before, target)
    kept = np.unique(target)
    new_index = np.full(n, -1, dtype=np.int64)
    new_index[kept] = np.arange(kept.shape[0])
    faces = new_index[target[faces]] if faces.size else faces
    good = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    faces = faces[good]
    if faces.shape[0]:
        _, first = np.unique(np.sort(faces, axis=1), axis=0, return_index=True)
        faces = faces[np.sort(first)]
    return points[kept], faces


def clean_mesh(mesh, distance_tolerance=1e-6):
    """``clean_mesh_raw`` on a mesh object; returns a new ``PolyData``
    (tfrt/mesh_tools.py:1041-1070)."""
    points, faces = clean_mesh_raw(mesh.points, mesh.triangles(), distance_tolerance)
    return PolyData(points, pack_faces(faces))


def planar_interpolated_remesh(input_mesh, base_mesh, range_axis=2, interp_fill_value=0.0,
                               flatten=True):
    """Re-mesh ``input_mesh`` on the (planar, regular) triangulation of ``base_mesh``: the
    height of the input vertices along ``range_axis`` is interpolated linearly over the other
    two coordinates and sampled at the base vertices (tfrt/mesh_tools.py:956-1031).  With
    ``flatten`` returns (flat copy of the base mesh, heights) - zero points plus initial
    parameters - otherwise the displaced copy."""
    from scipy.interpolate import griddata

    if range_axis not in (0, 1, 2):
        raise ValueError("planar_interpolated_remesh: axis must be in {0, 1, 2}.")
    domain = [a for a in (0, 1, 2) if a != range_axis]
    heights = griddata(input_mesh.points[:, domain], input_mesh.points[:, range_axis],
                       base_mesh.points[:, domain], fill_value=interp_fill_value)
    out = base_mesh.copy()
    out.points[:, range_axis] = 0.0 if flatten else heights
    return (out, heights) if flatten else out
