"""Glue between tests/scene_util.py scenes and the oracle (tests only)."""
import numpy as np
import torch

from oracle import tracer

F64 = torch.float64


def lens_oracle(scene, p_f=None, p_b=None, update_map_f=None, update_map_b=None):
    """Builds the oracle System for scene_util.lens_scene.  Returns (system, params, fields)."""
    tt = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    p_f = tt(scene["p_f"] if p_f is None else p_f).requires_grad_(True)
    p_b = tt(scene["p_b"] if p_b is None else p_b).requires_grad_(True)
    vec = tt(scene["vector"]).reshape(1, 3)
    v_f = tt(scene["zero_f"]) + p_f.reshape(-1, 1) * vec
    v_b = tt(scene["zero_b"]) + p_b.reshape(-1, 1) * vec
    front = tracer.faces_from_vertices(v_f, scene["faces_f"], update_map_f)
    back = tracer.faces_from_vertices(v_b, scene["faces_b"], update_map_b)
    for s in (front, back):
        n = s["xp"].shape[0]
        s["mat_in"] = torch.ones(n, dtype=torch.int64)
        s["mat_out"] = torch.zeros(n, dtype=torch.int64)
    optical = tracer.amalgamate([front, back])
    target = tracer.faces_from_vertices(tt(scene["target_verts"]), scene["target_faces"])
    system = tracer.System(
        3, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
        optical=optical, target=target)
    return system, (p_f, p_b), dict(front=front, back=back, target=target)


def source_dict(rays, wavelength, dtype=None, extra=None):
    """(6,N) block -> oracle ray set.  ``dtype=np.float32`` rounds the inputs first so the
    oracle sees exactly what a float32-state GPU trace sees."""
    rays = np.asarray(rays)
    if dtype is not None:
        rays = rays.astype(dtype).astype(np.float64)
    names = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
    d = {n: torch.tensor(rays[i], dtype=F64) for i, n in enumerate(names)}
    d["wavelength"] = torch.tensor(np.asarray(wavelength), dtype=F64)
    d["ray_id"] = torch.arange(rays.shape[1], dtype=F64)
    if extra:
        d.update(extra)
    return d


def block(rayset, dim=3):
    names = (("x_start", "y_start", "z_start", "x_end", "y_end", "z_end") if dim == 3 else
             ("x_start", "y_start", "x_end", "y_end"))
    if not rayset:
        return np.zeros((len(names), 0))
    return np.stack([rayset[n].detach().numpy() for n in names])
