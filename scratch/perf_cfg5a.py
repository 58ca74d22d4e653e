"""cfg5a: 3-D hex lens + ParametricCylindricalGuide(theta_res=64, z_res=64) + target
through the public API, N rays, float32 vs float16 vs float64 ray state; forward trace time and
the error of the reduced-precision states against float64 (same rays, same scene)."""
import sys, os, time, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import tfrt.boundaries as boundaries, tfrt.engine as engine, tfrt.materials as materials
import tfrt.operation as operation, tfrt.sources as sources, tfrt.mesh_tools as mt
import tfrt.distributions as distributions, tfrt.drawing as drawing
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
PASSES = 8

def build(ray_dtype):
    # light enters a collimating hex lens at z ~ 0 and is piped along +z by a tapered guide
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
    start = distributions.StaticUniformCircle(N, 0.05)
    distributions.BasePointTransformation(start, rotation=None, translation=(0, 0, -1.0)) \
        if False else None
    end = distributions.StaticUniformCircle(N, 0.42)
    src = sources.ManualSource(3)
    start.update(); end.update()
    sp, ep = start.points, end.points
    z0 = torch.full((N,), -1.0, dtype=torch.float64, device=sp.device)
    src["x_start"], src["y_start"], src["z_start"] = sp[:, 0], sp[:, 1], z0
    src["x_end"], src["y_end"], src["z_end"] = ep[:, 0], ep[:, 1], z0 + 1.2
    src["wavelength"] = torch.full((N,), float(drawing.YELLOW), dtype=torch.float64, device=sp.device)
    system = engine.OpticalSystem3D()
    system.optical = [front, back, guide]
    system.targets = [target]
    system.sources = [src]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(3, [operation.StandardReaction()], ray_dtype=ray_dtype,
                               simple_ray_inheritance={"wavelength"})
    eng.optical_system = system
    if os.environ.get("TFRT_COHERENT"):
        eng.coherent = {"0": False, "1": True}.get(os.environ["TFRT_COHERENT"], "auto")
    return eng, system

ref = None
for dt in (torch.float64, torch.float32, torch.float16):
    eng, system = build(dt)
    M = int(system._merged_face_verts.shape[0])
    for _ in range(2): eng.ray_trace(PASSES)
    torch.cuda.synchronize(); t = time.perf_counter(); K = 4
    for _ in range(K): eng.ray_trace(PASSES)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / K * 1e3
    fin = eng.finished_rays
    n_fin = fin["x_end"].shape[0]
    ids = eng.last_trace["finished_id"].long()
    xy = torch.stack([fin["x_end"], fin["y_end"]]).double()
    line = (f"{str(dt)[6:]:8s} N={N} M={M} passes={PASSES} coherent={eng.coherent}: {ms:7.2f} ms/trace, "
            f"{eng.last_trace['n_tests']/ms*1e3:.3e} tests/s, finished {n_fin}, left over {eng.last_trace.get('left_over')}")
    if ref is None:
        ref = (ids, xy)
    else:
        a, b = ids.cpu().numpy(), ref[0].cpu().numpy()
        common, ia, ib = np.intersect1d(a, b, return_indices=True)
        err = (xy[:, torch.as_tensor(ia, device=xy.device)] - ref[1][:, torch.as_tensor(ib, device=xy.device)]).abs()
        line += (f", {common.shape[0]} of {b.shape[0]} finished rays in common, |end - f64| on them: "
                 f"median {err.median().item():.2e}, 99.9 % {torch.quantile(err.flatten()[::7].float(), 0.999).item():.2e}, max {err.max().item():.2e}")
    print(line, flush=True)
