"""TEST-ONLY placeholder so that tfrt/boundaries.py and tfrt/mesh_tools.py (which import pyvista at
module level) can be loaded by tests/golden/make_reference_host_golden.py.  ``PolyData`` carries
what the parametric boundaries read from a mesh: points, the flat faces array, copy()."""
import numpy as np


class PolyData:
    def __init__(self, points=None, faces=None):
        if isinstance(points, PolyData):
            points, faces = points.points, points.faces
        self.points = None if points is None else np.array(points, dtype=np.float64)
        self.faces = None if faces is None else np.array(faces, dtype=np.int64).reshape(-1)

    def copy(self):
        return PolyData(self.points, self.faces)

    @property
    def n_faces(self):
        return 0 if self.faces is None else self.faces.size // 4


def read(_filename):
    raise NotImplementedError("pyvista.read is not available (test shim)")
