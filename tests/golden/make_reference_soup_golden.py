#!/usr/bin/env python3
"""
Golden vectors of the 3-D trace on adversarial scenes, produced by EXECUTING the reference's own
engine (tfrt/engine.py, tfrt/operation.py StandardReaction, tfrt/geometry.py) in the build
container under the TensorFlow stand-in of tests/tf_shim.  Writes tests/golden/reference_soup3d.npz.

Scenes: the triangle soups of tests/test_gpu_stress.py::_soup (random scale and offset, grazing
rays, a third of the faces in one plane so that coplanar faces tie in ray_u, 10 % stops, 10 %
targets) -- exactly the inputs on which the discrete decisions (tf.argmin's first index,
engine.py:1149; the epsilon windows, engine.py:1130-1140) are delicate.  Every ray class the
engine can compile is stored: active, finished, dead (with dead_ray_length set on one run,
engine.py:1962-1990), stopped, and the unfinished rays left after the last pass.

Gradients ("plain" runs): d loss / d face vertices by torch.autograd over the reference's op
sequence, loss = a scalar of the finished, active-history and stopped rays scaled by the scene's
length scale (soup_loss below).  Mirrors (n_in = 0) and total internal reflection take the
reflect branch of snells_law_3D (geometry.py:735-747), whose unselected sqrt sees a safe radicand:
unlike the 2-D form these gradients are finite; what non-finite entries there are come from
degenerate (grazing / parallel) pairs and are recorded as they are.

The reference takes refractive indices through material indices (engine.py:1166-1197), so each
face gets mat_in / mat_out into a list of six constant-index materials (among them index 0 =
reflective, materials.py) instead of the stress test's per-face random indices.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, os.path.join(TESTS, "tf_shim"))
if not os.path.exists("/root/reference/tfrt/engine.py"):
    raise SystemExit("the reference is not present here: fixtures can only be made in the build container")
sys.path.insert(0, "/root/reference")

import tfrt.engine as ref_engine        # noqa: E402  (the reference's modules)
import tfrt.operation as ref_operation  # noqa: E402
from make_reference_trace_golden import FieldSet, faces_from_vertices  # noqa: E402

F64 = torch.float64
GEO = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
N_MATERIAL = (0.0, 1.0, 1.21, 1.37, 1.52, 1.69)       # 0 = mirror (geometry.py:735-741)
SEEDS = (35, 2, 16, 25)
PASSES = 3


def soup(seed):
    """tests/test_gpu_stress.py::_soup with material indices instead of per-face random n."""
    rng = np.random.default_rng(5000 + seed)
    n_faces = int(rng.choice([64, 97, 300, 640]))
    n_rays = int(rng.choice([50, 700, 2500]))
    scale = 10 ** rng.uniform(-3, 3)
    offset = rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(0, 2.5) * (rng.random() < 0.5)
    centre = rng.uniform(-1, 1, (n_faces, 1, 3))
    size = 10 ** rng.uniform(-2.5, -0.2, (n_faces, 1, 1))
    tri = (centre + size * rng.standard_normal((n_faces, 3, 3))) * scale + offset
    if rng.random() < 0.5 or seed == 35:
        tri[: n_faces // 3, :, 2] = offset[2] + 0.1 * scale
    cat = np.zeros(n_faces, dtype=np.int64)
    cat[int(0.8 * n_faces):int(0.9 * n_faces)] = 1
    cat[int(0.9 * n_faces):] = 2
    mat_in = rng.integers(1, len(N_MATERIAL), n_faces)
    mat_out = rng.integers(1, len(N_MATERIAL), n_faces)
    mat_in[rng.random(n_faces) < 0.08] = 0               # some mirrors
    s = rng.uniform(-1.5, 1.5, (3, n_rays)) * scale + offset[:, None]
    d = rng.standard_normal((3, n_rays))
    if rng.random() < 0.5:
        d[2] *= 1e-3
    e = s + d * scale * 10 ** rng.uniform(-2, 0.5)
    return dict(P=tri.reshape(n_faces, 9), cat=cat, mat_in=mat_in, mat_out=mat_out,
                rays=np.concatenate([s, e]), L=float(scale))


def soup_loss(fin, act, stp, L):
    """fin / act / stp: 6 x n blocks (or None).  Used by the tests as well."""
    loss = 0.0
    if fin is not None:
        loss = loss + ((fin[3] / L) ** 2).sum() + 0.5 * (fin[4] * fin[2]).sum() / L ** 2
    if act is not None:
        loss = loss + 0.3 * act[5].sum() / L
    if stp is not None:
        loss = loss + 0.2 * stp[4].sum() / L
    return loss


def reference_trace(sc, dead_ray_length, grads=False):
    tt = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    P_all = tt(sc["P"]).requires_grad_(grads)

    def sub(mask, optical):
        verts = P_all[torch.tensor(mask)].reshape(-1, 3)
        fs = faces_from_vertices(verts, np.arange(verts.shape[0]).reshape(-1, 3))
        if optical:
            fs["mat_in"] = torch.tensor(sc["mat_in"][mask], dtype=torch.int64)
            fs["mat_out"] = torch.tensor(sc["mat_out"][mask], dtype=torch.int64)
        fs["face_index"] = tt(np.nonzero(mask)[0])
        return fs

    source = FieldSet({k: tt(sc["rays"][i]) for i, k in enumerate(GEO)})
    n_rays = sc["rays"].shape[1]
    source["wavelength"] = torch.full((n_rays,), 550.0, dtype=F64)
    source["ray_id"] = torch.arange(n_rays, dtype=F64)
    system = ref_engine.OpticalSystem3D()
    system.optical = [sub(sc["cat"] == 0, True)]
    system.stops = [sub(sc["cat"] == 1, False)]
    system.targets = [sub(sc["cat"] == 2, False)]
    system.sources = [source]
    system.materials = [{"n": (lambda wl, v=v: v * torch.ones_like(wl))} for v in N_MATERIAL]
    system.update()
    eng = ref_engine.OpticalEngine(
        3, [ref_operation.StandardReaction()], compile_dead_rays=True, compile_stopped_rays=True,
        dead_ray_length=dead_ray_length, new_ray_length=sc["L"],
        simple_ray_inheritance={"wavelength", "ray_id"})
    eng.optical_system = system
    eng.validate_system()
    # ray_trace (engine.py:2311-2330) written out with the engine's public single_pass, because
    # ray_trace drops the rays still travelling after the last pass and they are compared too
    eng.clear_ray_history()
    travelling = system._amalgamated_sources.copy()
    for _ in range(PASSES):
        result = eng.single_pass(travelling)
        if not bool(result):
            travelling = {}
            break
        travelling = result
    out = {}
    if grads:
        blk = lambda rs: torch.stack([rs[g] for g in GEO]) if bool(rs) else None
        loss = soup_loss(blk(eng.finished_rays), blk(eng.active_rays), blk(eng.stopped_rays), sc["L"])
        (g_P,) = torch.autograd.grad(loss, [P_all])
        out["loss"], out["grad_P"] = np.float64(loss.item()), g_P.numpy()
    for cls, rs in (("finished", eng.finished_rays), ("active", eng.active_rays),
                    ("dead", eng.dead_rays), ("stopped", eng.stopped_rays),
                    ("unfinished", travelling)):
        n = rs["x_start"].shape[0] if "x_start" in rs.keys() else 0
        out[cls] = (torch.stack([rs[g] for g in GEO]).detach().numpy() if n
                    else np.zeros((6, 0)))
        out[cls + "_id"] = (rs["ray_id"].detach().numpy().astype(np.int64) if n
                            else np.zeros(0, dtype=np.int64))
    return out


def main():
    out = {"seeds": np.array(SEEDS), "passes": np.int64(PASSES), "n_material": np.array(N_MATERIAL)}
    for seed in SEEDS:
        sc = soup(seed)
        for k, v in sc.items():
            out[f"s{seed}__{k}"] = np.asarray(v)
        for tag, dl in (("plain", None), ("deadlen", 2.5 * sc["L"])):
            res = reference_trace(sc, dl, grads=tag == "plain")
            for k, v in res.items():
                out[f"s{seed}__{tag}__{k}"] = v
            print(seed, tag, {k: v.shape[-1] for k, v in res.items()
                              if not k.endswith("_id") and k not in ("loss", "grad_P")},
                  "" if "grad_P" not in res else
                  f"grad_P: {int((~np.isfinite(res['grad_P'])).sum())} non-finite of "
                  f"{res['grad_P'].size}, {int((res['grad_P'] != 0).any(axis=1).sum())} faces touched")
        out[f"s{seed}__dead_ray_length"] = np.float64(2.5 * sc["L"])
    np.savez_compressed(os.path.join(HERE, "reference_soup3d.npz"), **out)


if __name__ == "__main__":
    main()
