"""
TEST-ONLY stand-ins for the HIP-backed ops the host logic calls (face build, trace, gradient
processing, CSR product), built on the oracle, so
the Python host layer (engine field assembly, inheritance, optimizer, ray sharding, gradient
all-reduce) can be exercised in a container without a GPU.  Installed by the ``cpu_backend``
fixture via monkeypatch; never imported by the product package.
"""
import numpy as np
import torch

from oracle import tracer

_COLS = ("xp", "yp", "zp", "x1", "y1", "z1", "x2", "y2", "z2")
_GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


def build_faces(vertices, faces, update_mask=None):
    f = tracer.faces_from_vertices(
        vertices.double(), faces.long(),
        None if update_mask is None else update_mask.bool())
    return torch.stack([f[c] for c in _COLS], dim=1), f["norm"]


def trace3d(src, face_verts, scene, max_passes, new_ray_length=1.0, dead_ray_length=None,
            flags=3, predicted_counts=None, perm=None, ray_slot=None, lazy=False):
    assert perm is None          # (a coherent order is a device matter: CPU blocks are never sorted)
    n = src.shape[1]
    fv = face_verts.double()
    fields = {c: fv[:, i] for i, c in enumerate(_COLS)}
    cross = torch.linalg.cross(fv[:, 3:6] - fv[:, 0:3], fv[:, 6:9] - fv[:, 3:6], dim=1)
    fields["norm"] = cross / torch.linalg.norm(cross, dim=1, keepdim=True)
    cat = scene.catagory.long()
    sets = {}
    for name, c in (("optical", 0), ("stop", 1), ("target", 2)):
        m = cat == c
        if bool(m.any()):
            s = {k: v[m] for k, v in fields.items()}
            s["face_index"] = torch.nonzero(m).reshape(-1).double()
            if c == 0:
                if scene.mat_in is not None:
                    s["mat_in"] = scene.mat_in.long()[m]
                    s["mat_out"] = scene.mat_out.long()[m]
                else:
                    s["n_in"] = scene.n_in[m]
                    s["n_out"] = scene.n_out[m]
            sets[name] = s
    index_mode = scene.mat_in is not None
    mats = []
    if index_mode:
        for row in scene.n_table:
            if getattr(scene, "n_table_uniform", False):      # one column for every ray
                mats.append(lambda rid, row=row: row[0].expand(rid.shape[0]))
            else:
                mats.append(lambda rid, row=row: row[rid.long()])
    ie, se, rse = scene.eps
    system = tracer.System(3, materials=mats, intersect_epsilion=ie, size_epsilion=se,
                           ray_start_epsilion=rse, **sets)
    rays = {g: src[i].double() for i, g in enumerate(_GEO3)}
    rays["wavelength"] = torch.arange(n, dtype=torch.float64)  # carries the source-ray id
    fl = dict(compile_active_rays=bool(flags & 1), compile_finished_rays=bool(flags & 2),
              compile_stopped_rays=bool(flags & 4), compile_dead_rays=bool(flags & 8),
              dead_ray_length=dead_ray_length)
    history = {"active": [], "finished": [], "stopped": [], "dead": []}
    counts = np.zeros((max_passes, 8), dtype=np.int32)
    n_tests = 0
    m_faces = fv.shape[0]
    for p in range(max_passes):
        before = {k: len(v) for k, v in history.items()}
        n_in = rays["x_start"].shape[0] if rays else 0
        if not rays:
            break
        n_tests += n_in * m_faces
        rays, _ = tracer.single_pass(system, rays, history, flags=fl,
                                     new_ray_length=new_ray_length, inherit=("wavelength",),
                                     index_type="index" if index_mode else "value")
        counts[p, 0] = rays["x_start"].shape[0] if rays else 0
        for k, cls in ((1, "finished"), (2, "stopped"), (3, "dead")):
            if len(history[cls]) > before[cls]:
                counts[p, k] = history[cls][-1]["x_start"].shape[0]
    out = {"counts": counts, "n_tests": n_tests, "raw_counts": counts.reshape(-1)}
    for cls, flag in (("finished", 2), ("active", 1), ("stopped", 4), ("dead", 8)):
        if not (flags & flag):
            continue
        h = tracer.amalgamate(history[cls])
        if h:
            out[cls] = torch.stack([h[g] for g in _GEO3]).to(src.dtype)
            out[cls + "_id"] = h["wavelength"].to(torch.int32)
        else:
            out[cls] = torch.zeros((6, 0), dtype=src.dtype)
            out[cls + "_id"] = torch.zeros(0, dtype=torch.int32)
        out[cls + "_face"] = torch.full_like(out[cls + "_id"], -1)
    if rays:
        out["unfinished"] = torch.stack([rays[g] for g in _GEO3]).to(src.dtype)
        out["unfinished_id"] = rays["wavelength"].to(torch.int32)
    else:
        out["unfinished"] = torch.zeros((6, 0), dtype=src.dtype)
        out["unfinished_id"] = torch.zeros(0, dtype=torch.int32)
    return out


def sgd_process(grad, scale, clip, param=None, sgd_learning_rate=0.0):
    """optimizer.py:223-247 (+ :316 with ``param``) as eager torch ops (what tfrt_sgd_process does
    on the device)."""
    out = torch.where(torch.isfinite(grad), grad, torch.zeros_like(grad)) * scale
    out = torch.clamp(out, -clip, clip)
    if param is not None:
        with torch.no_grad():
            param.add_(out, alpha=-sgd_learning_rate)
    return out


class CsrMatrix:
    """Dense stand-in of ops.CsrMatrix (optimizer.py:250-255, 277-282 multiply dense matrices)."""

    def __init__(self, matrix, device):
        m = matrix if isinstance(matrix, torch.Tensor) else torch.as_tensor(np.asarray(matrix))
        if m.layout != torch.strided:
            m = m.to_dense()
        self._m = m.to(device=device, dtype=torch.float64)
        self.shape = tuple(self._m.shape)

    def matvec(self, x):
        y = torch.matmul(self._m, x.reshape(-1, 1).to(torch.float64)).reshape(-1)
        return y.reshape(x.shape).to(x.dtype) if self.shape[0] == self.shape[1] else y


def install(monkeypatch):
    import tensorflowraytrace_amd as tfa
    from tensorflowraytrace_amd import ops
    tfa.set_device("cpu")
    monkeypatch.setattr(ops, "build_faces", build_faces)
    monkeypatch.setattr(ops, "trace3d", trace3d)
    monkeypatch.setattr(ops, "sgd_process", sgd_process)
    monkeypatch.setattr(ops, "CsrMatrix", CsrMatrix)
