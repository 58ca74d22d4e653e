"""Scenes of the other BASELINE configurations: cfg5a -- hex lens +
ParametricCylindricalGuide + target, 4M rays, 8 passes, through the public API -- and cfg5b -- 2-D,
64 arcs + 256 segments, 4M rays.  Shared by the full-size GPU tests and bench.py --config."""
import numpy as np
import torch

N_5A, PASSES_5A = 4_000_000, 8


def _build_5a(ray_dtype, n_rays=N_5A, accelerate="auto", ray_shard=None, compile_all=True,
              coherent="auto"):
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.drawing as drawing
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources

    def surface(k, flip, sign, z):
        zp = mt.hexagonal_mesh(0.45, k)                 # in the x-y plane
        zp.points[:, 2] = z
        r2 = (zp.points[:, 0] ** 2 + zp.points[:, 1] ** 2) / 0.45 ** 2
        return boundaries.ParametricTriangleBoundary(
            zp, boundaries.FromVectorVG((0, 0, 1)), flip_norm=flip,
            initial_parameters=sign * (0.02 + 0.05 * (1 - r2)),
            material_dict={"mat_in": 1, "mat_out": 0})

    front, back = surface(24, True, -1.0, 0.3), surface(24, False, +1.0, 0.5)
    guide = boundaries.ParametricCylindricalGuide(
        (0, 0, 1.0), (0, 0, 7.0), 0.5, theta_res=64, z_res=64, initial_taper=(0.0, 0.15),
        material_dict={"mat_in": 1, "mat_out": 0})
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(0, 0, 6.9), direction=(0, 0, 1), i_size=3, j_size=3))
    start = distributions.StaticUniformCircle(n_rays, 0.05)
    end = distributions.StaticUniformCircle(n_rays, 0.42)
    start.update()
    end.update()
    sp, ep = start.points, end.points
    z0 = torch.full((n_rays,), -1.0, dtype=torch.float64, device=sp.device)
    src = sources.ManualSource(3)
    src["x_start"], src["y_start"], src["z_start"] = sp[:, 0], sp[:, 1], z0
    src["x_end"], src["y_end"], src["z_end"] = ep[:, 0], ep[:, 1], z0 + 1.2
    src["wavelength"] = torch.full((n_rays,), float(drawing.YELLOW), dtype=torch.float64,
                                   device=sp.device)
    system = engine.OpticalSystem3D()
    system.optical = [front, back, guide]
    system.targets = [target]
    system.sources = [src]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(
        3, [operation.StandardReaction()], ray_dtype=ray_dtype, accelerate=accelerate,
        compile_dead_rays=compile_all, compile_stopped_rays=compile_all,
        compile_active_rays=compile_all, simple_ray_inheritance={"wavelength"},
        ray_shard=ray_shard, coherent=coherent)
    eng.optical_system = system
    return eng, system, (front, back, guide, target)


def _scene_5b(n_rays, seed=0):
    import math
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    na = 64
    xc = np.linspace(-16, 16, na)
    xs = np.linspace(-18, 18, 256)
    ys = 6.0 + 0.3 * np.sin(xs)
    sets = {
        "optical_arcs": dict(x_center=t(xc), y_center=t(np.full(na, 3.0)),
                             angle_start=t(np.full(na, -math.pi + 0.3)), angle_end=t(np.full(na, -0.3)),
                             radius=t(np.full(na, 0.6)), mat_in=torch.ones(na, dtype=torch.int64),
                             mat_out=torch.zeros(na, dtype=torch.int64)),
        "optical_segments": dict(x_start=t(xs[:-1]), y_start=t(ys[:-1]), x_end=t(xs[1:]), y_end=t(ys[1:]),
                                 mat_in=torch.full((255,), 2, dtype=torch.int64),
                                 mat_out=torch.zeros(255, dtype=torch.int64)),
        "target_segments": dict(x_start=t([20.0]), y_start=t([-1.0]), x_end=t([20.0]), y_end=t([9.0])),
    }
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0.3 * math.pi, 0.7 * math.pi, n_rays)
    x0 = rng.uniform(-15, 15, n_rays)
    rays = np.stack([x0, np.zeros(n_rays), x0 + np.cos(ang), np.sin(ang)])
    wl = np.full(n_rays, 550.0)
    return sets, rays, wl


def _oracle_5a(parts):
    from oracle import tracer
    front, back, guide, target = parts
    cpu = lambda t: t.detach().cpu()
    sets = []
    for b in (front, back, guide):
        f = tracer.faces_from_vertices(cpu(b.vertices), b.faces[:, 1:])
        n = f["xp"].shape[0]
        f["mat_in"] = torch.ones(n, dtype=torch.int64)
        f["mat_out"] = torch.zeros(n, dtype=torch.int64)
        sets.append(f)
    tgt = tracer.faces_from_vertices(cpu(target.vertices), target.faces[:, 1:])
    return tracer.System(3, materials=[tracer.MATERIALS["vacuum"], tracer.MATERIALS["acrylic"]],
                         optical=tracer.amalgamate(sets), target=tgt)


