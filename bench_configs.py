"""bench.py --config {cfg2, cfg3, cfg5a, cfg5b}: the other BASELINE.json configurations, measured
the same way as the headline (cfg4, bench.py itself): warm-up, K timed steps between device
synchronisations, one JSON line of the same shape.

  cfg2   3-D aperture source, 100,000 rays x 974 faces (two H(9) surfaces + target), 5 passes,
         forward only: optical_system.update() + OpticalEngine.ray_trace(5) per step
  cfg3   the same scene with the backward pass: SGD_Optimizer.single_step, trace_depth 5
  cfg5a  hex lens + ParametricCylindricalGuide(64, 64) + target = 15,106 faces, 4,000,000 rays,
         OpticalEngine.ray_trace(8), float32 ray state (float16 / float64 as side fields)
  cfg5b  2-D: 64 arcs + 256 segments, 4,000,000 rays, 4 passes, forward + reverse sweep (ops layer)

A step's algorithmic HBM bytes are SURVEY.md 8d's: 64 B per ray entering a pass (+ 116 B with the
reverse sweep) + 48 B per primitive and pass; `roofline` relates them to the step time.  The
`cpu_baseline` is the oracle (reference algorithm, dense float64 torch ops) on a bounded sample of
the same workload, on the host cores.
"""
import json
import os
import time

import numpy as np
import torch

PEAK_HBM_GBPS = 8000.0
B_FWD, B_BWD, B_PRIM = 64, 116, 48

WORKLOADS = {
    "cfg2": "cfg2: 3-D aperture source 100k rays x 974-face lens (2 x H(9) + target), 5 passes, "
            "forward only (update + ray_trace)",
    "cfg3": "cfg3: cfg2's scene with the backward pass (SGD_Optimizer.single_step, trace_depth 5)",
    "cfg5a": "cfg5a: hex lens + cylindrical guide (64 x 64) + target = 15,106 faces, 4M rays, "
             "ray_trace(8), forward",
    "cfg5b": "cfg5b: 2-D, 64 arcs + 256 segments, 4M rays, 4 passes, forward + reverse sweep",
}


def _cores():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def _timed(step, warmup, steps):
    """warm-up, then EXACTLY `steps` steps between device synchronisations"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0), steps


def _cpu_lens(k_front, k_back, passes, backward, seconds):
    import scene_util
    import oracle_util
    from oracle import tracer
    cores = _cores()
    torch.set_num_threads(cores)
    chunk, warm = 1024, 1
    scene = scene_util.lens_scene(200_000, k_front=k_front, k_back=k_back)
    tests = rays_done = k = 0
    t0 = None
    while (k + 1) * chunk <= scene["rays"].shape[1]:
        if k == warm:
            t0 = time.time()
        lo, hi = k * chunk, (k + 1) * chunk
        k += 1
        system, (p_f, p_b), _ = oracle_util.lens_oracle(scene)
        ref = tracer.ray_trace(
            system, oracle_util.source_dict(scene["rays"][:, lo:hi], scene["wavelength"][lo:hi]),
            max_iterations=passes, inherit=("wavelength", "ray_id"), chunk=chunk)
        if backward:
            fin = ref["finished"]
            goal = torch.tensor(scene["goal"][lo:hi], dtype=torch.float64)[fin["ray_id"].long()]
            err = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
            torch.autograd.grad(err, [p_f, p_b])
        if t0 is None:
            continue
        m = system.merged["xp"].shape[0]
        tests += (chunk + (int(ref["active"]["x_start"].shape[0]) if ref["active"] else 0)) * m
        rays_done += chunk
        if time.time() - t0 > seconds:
            break
    dt = time.time() - t0
    return {"value": tests / dt, "unit": "tests/s", "cores": cores, "kind": "port",
            "sample": f"{rays_done} consecutive source rays of a 200,000-ray instance of the scene, "
                      f"{passes} passes, oracle (torch-CPU float64, dense (M,N) temporaries) "
                      f"forward{' + autograd' if backward else ''} in {chunk}-ray chunks; {dt:.1f} s wall"}


def _line(name, dt, steps, args, tests, n_active, M, backward, extra_config, cpu, dtype):
    P = len(n_active)
    step_bytes = sum(n_active) * (B_FWD + (B_BWD if backward else 0)) + P * M * B_PRIM
    ms = dt / steps * 1e3
    gbps = step_bytes / (ms * 1e-3) / 1e9
    line = {
        "metric": "ray-surface intersection tests/sec (" + ("fwd+bwd" if backward else "fwd") + ")",
        "value": tests / dt, "unit": "tests/s", "n_gpus": 1, "steps": steps,
        "warmup": args.warmup, "ms_per_step": ms, "timed_region_s": dt,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": dict({"workload": WORKLOADS[name], "faces": M, "passes": P,
                        "rays_entering_each_pass": n_active}, **extra_config),
        "roofline": {
            "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": gbps / PEAK_HBM_GBPS, "traffic": None,
            "scope": "whole step: SURVEY 8d algorithmic bytes (%.1f MB: 64 B%s per ray entering a "
                     "pass + 48 B per primitive and pass) / ms_per_step"
                     % (step_bytes / 1e6, " + 116 B" if backward else ""),
        },
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


def _kernel_rows(lib, n_active, M, P):
    """per-launch times of the hot 3-D kernels (HIP events, tfrt_profile_*) of the eager steps run
    since profiling was enabled"""
    import ctypes
    buf = (ctypes.c_float * 8192)()
    rows = []
    n_fwd = float(np.mean(n_active))
    for kind, name, alg in ((0, "intersect", n_fwd * B_FWD + M * B_PRIM), (1, "react", n_fwd * B_FWD),
                            (2, "backward", n_fwd * B_BWD), (3, "accumulate", None)):
        nrec = lib.tfrt_profile_read_kind(kind, buf, 8192)
        v = np.asarray([buf[i] for i in range(max(nrec, 0))], dtype=np.float64)
        if not v.size:
            continue
        row = {"kind": name, "launches_timed": int(v.size), "avg_ms": float(v.mean())}
        if alg is not None:
            row["algorithmic_bytes"] = alg
            row["hbm_frac"] = alg / (float(v.mean()) * 1e-3) / 1e9 / PEAK_HBM_GBPS
        rows.append(row)
    return rows


def run(name, args):
    import bench
    import tensorflowraytrace_amd as tfa
    from tensorflowraytrace_amd import _lib, ops
    import tfrt.optimizer as optimizer
    torch.cuda.set_device(0)
    tfa.set_device("cuda:0")
    lib = _lib.lib()
    cpu = None
    if name in ("cfg2", "cfg3"):
        backward = name == "cfg3"
        eng, system, params = bench.build_scene(100_000, 9, 9, torch.float32)
        if backward:
            opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=5,
                                          learning_rate=1e-6, grad_clip=1e-3)
            opt.suppress_warnings = True
            step = lambda: opt.single_step(None)
        else:
            def step():
                system.update()
                eng.ray_trace(5)
        dt, steps = _timed(step, args.warmup, args.steps)
        counts = eng.last_trace["counts"]
        M = int(system._merged_face_verts.shape[0])
        n_active = [int(c[:4].sum()) for c in counts]
        tests = float(sum(n_active)) * M * steps
        # per-kernel launch times: the same steps once more, eagerly, with the event hooks on
        lib.tfrt_profile_enable(1)
        if backward:
            opt2 = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=5,
                                           learning_rate=1e-6, grad_clip=1e-3, graph=False)
            opt2.suppress_warnings = True
            for _ in range(10):
                opt2.single_step(None)
        else:
            for _ in range(10):
                step()
        torch.cuda.synchronize()
        rows = _kernel_rows(lib, n_active, M, len(n_active))
        lib.tfrt_profile_enable(0)
        if not args.no_cpu_baseline:
            cpu = _cpu_lens(9, 9, 5, backward, args.cpu_seconds)
        extra = {"rays": 100_000, "step": "SGD_Optimizer.single_step (fused launch sequence, HIP-graph "
                 "replay)" if backward else "optical_system.update() + OpticalEngine.ray_trace(5), "
                 "ray sets cut (one host read of the counts per step)",
                 "ordered": getattr(eng, "_order_cache", None) is not None}
        line = _line(name, dt, steps, args, tests, n_active, M, backward, extra, cpu,
                     "f32 ray state, f32 filter + f64 decisions")
        line["roofline"]["kernels"] = rows
    elif name == "cfg5a":
        import scene_configs as sc5
        legs = {}
        for tag, dt_ in (("f32", torch.float32), ("f16", torch.float16), ("f64", torch.float64)):
            eng, system, parts = sc5._build_5a(dt_, compile_all=False)
            t, steps = _timed(lambda: eng.ray_trace(sc5.PASSES_5A), min(args.warmup, 3),
                              min(args.steps, 10))
            counts = eng.last_trace["counts"]
            M = int(system._merged_face_verts.shape[0])
            n_active = [int(c[:4].sum()) for c in counts]
            legs[tag] = dict(dt=t, steps=steps, n_active=n_active, M=M,
                             ordered=eng._trace_perm is not None,
                             left_over=int(eng.last_trace.get("left_over", 0)))
            if tag == "f32":
                lib.tfrt_profile_enable(1)
                for _ in range(3):
                    eng.ray_trace(sc5.PASSES_5A)
                torch.cuda.synchronize()
                rows = _kernel_rows(lib, n_active, M, len(n_active))
                lib.tfrt_profile_enable(0)
                eng_nat, *_ = sc5._build_5a(dt_, compile_all=False, coherent=False)
                tn, sn = _timed(lambda: eng_nat.ray_trace(sc5.PASSES_5A), 2, 5)
                legs["f32_natural_order"] = dict(dt=tn, steps=sn)
                if not args.no_cpu_baseline:
                    cpu = _cpu_5a(parts, eng, args.cpu_seconds)
            del eng, system
            torch.cuda.empty_cache()
        m = legs["f32"]
        tests = float(sum(m["n_active"])) * m["M"] * m["steps"]
        extra = {"rays": sc5.N_5A, "step": "OpticalEngine.ray_trace(8) through the public API",
                 "ordered": m["ordered"], "wavefronts_left_to_the_grouped_kernel": m["left_over"],
                 "other_ray_states_ms": {k: v["dt"] / v["steps"] * 1e3 for k, v in legs.items()
                                         if k != "f32"}}
        line = _line(name, m["dt"], m["steps"], args, tests, m["n_active"], m["M"], False, extra, cpu,
                     "f32 ray state, f32 filter + f64 decisions")
        line["roofline"]["kernels"] = rows
    elif name == "cfg5b":
        import scene_configs as sc5
        import test_gpu_trace2d as t2
        N, P = 4_000_000, 4
        sets, rays, wl = sc5._scene_5b(N)
        scene, seg, arc = t2._gpu_scene(sets, wl, requires_grad=True)
        src = torch.tensor(rays, dtype=torch.float32, device="cuda:0")
        geo = [seg["geo"], arc["geo"]]
        last = {}

        def step():
            out = ops.trace2d(src, scene, P, flags=_lib.COMPILE_FINISHED)
            loss = (out["finished"][3].double() ** 2).sum()
            torch.autograd.grad(loss, geo)
            last["out"] = out
        dt, steps = _timed(step, min(args.warmup, 3), min(args.steps, 20))
        counts = last["out"]["counts"]
        n_active = [int(c[:4].sum()) for c in counts]
        M = 320
        tests = float(sum(n_active)) * M * steps
        if not args.no_cpu_baseline:
            cpu = _cpu_5b(sets, rays, wl, P, args.cpu_seconds)
        extra = {"rays": N, "step": "ops.trace2d forward + reverse sweep to the segment / arc "
                                    "geometry (one host read of the counts per step)"}
        line = _line(name, dt, steps, args, tests, n_active, M, True, extra, cpu,
                     "f32 ray state, f32 filter + f64 decisions")
    else:
        raise SystemExit(f"unknown --config {name}")
    print(json.dumps(line))


def _cpu_5a(parts, eng, seconds):
    import scene_configs as sc5
    from oracle import tracer
    cores = _cores()
    torch.set_num_threads(cores)
    system = sc5._oracle_5a(parts)
    src = eng.optical_system._amalgamated_sources
    names = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "wavelength")
    chunk, tests, done, k = 256, 0, 0, 0
    M = system.merged["xp"].shape[0]
    t0 = None
    rng = np.random.default_rng(3)
    while True:
        if k == 1:
            t0 = time.time()
        pick = torch.as_tensor(np.sort(rng.choice(sc5.N_5A, chunk, replace=False)), device="cuda")
        k += 1
        rays = {f: src[f][pick].detach().cpu().double() for f in names}
        rays["ray_id"] = torch.arange(chunk)
        ref = tracer.ray_trace(system, rays, max_iterations=sc5.PASSES_5A,
                               inherit=("wavelength", "ray_id"), chunk=chunk)
        if t0 is None:
            continue
        tests += (chunk + (int(ref["active"]["x_start"].shape[0]) if ref["active"] else 0)) * M
        done += chunk
        if time.time() - t0 > seconds:
            break
    dt = time.time() - t0
    return {"value": tests / dt, "unit": "tests/s", "cores": cores, "kind": "port",
            "sample": f"{done} source rays drawn at random from the 4M, 15,106 faces, 8 passes, oracle "
                      f"forward (torch-CPU float64, dense (M,N) temporaries) in {chunk}-ray chunks; "
                      f"{dt:.1f} s wall"}


def _cpu_5b(sets, rays, wl, P, seconds):
    import test_gpu_trace2d as t2
    from oracle import tracer
    cores = _cores()
    torch.set_num_threads(cores)
    system = t2._oracle_system(sets)
    chunk, tests, done, k = 8192, 0, 0, 0
    t0 = None
    while (k + 1) * chunk <= rays.shape[1]:
        if k == 1:
            t0 = time.time()
        lo, hi = k * chunk, (k + 1) * chunk
        k += 1
        ref = tracer.ray_trace(system, t2._src2(rays[:, lo:hi], wl[lo:hi], True), max_iterations=P,
                               inherit=("wavelength", "ray_id"))
        if t0 is None:
            continue
        tests += (chunk + (int(ref["active"]["x_start"].shape[0]) if ref["active"] else 0)) * 320
        done += chunk
        if time.time() - t0 > seconds:
            break
    dt = time.time() - t0
    return {"value": tests / dt, "unit": "tests/s", "cores": cores, "kind": "port",
            "sample": f"{done} consecutive source rays x 320 primitives, {P} passes, oracle forward "
                      f"(torch-CPU float64, dense (M,N) temporaries); {dt:.1f} s wall"}
