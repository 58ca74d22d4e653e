#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the tfrt hot path on MI355X.

Workload (BASELINE.json configs[3], SURVEY.md section 8d "cfg4"): 1,000,000 aperture-source
rays x a two-surface parametric acrylic lens (front hex mesh H(41) = 10,086 faces, back
H(9) = 486 faces) + a 2-triangle target = 10,574 merged faces; one step = one
``SGD_Optimizer.single_step`` with ``trace_depth=3``: constraints + parameters->faces,
3-pass trace, error function, hand-derived backward, gradient processing and SGD update.
Synthetic, deterministic inputs (golden-spiral source points).

Metric: ray-surface intersection tests/s (fwd+bwd), tests = sum over passes of
N_active(pass) x M_merged (counted by the kernels).  With N GPUs the source rays are sharded
contiguously over the ranks and the per-step parameter gradients are summed with one RCCL
all-reduce.  Default is weak scaling: every GPU traces 1M rays of a global source of N x 1M
(value = tests of all ranks / max-over-ranks time); the line also carries `strong_scaling`, the
N=1 workload itself split over the ranks (`--scaling strong` makes that the headline).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E spec peak (6290 measured copy)
FLOPS_PER_TEST = 45        # SURVEY.md section 8d algorithmic cost model (Moeller-Trumbore)
BYTES_PER_RAY_FWD = 64     # 32 B read + 32 B written per active ray per pass
BYTES_PER_FACE = 48


def build_scene(n_rays, k_front, k_back, ray_dtype, accelerate="auto"):
    import tensorflowraytrace_amd as tfa
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.drawing as drawing
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources

    start_points = distributions.StaticUniformCircle(n_rays, 0.2)
    distributions.BasePointTransformation(start_points, translation=(-10, 0, 0))
    end_points = distributions.StaticUniformCircle(n_rays, 0.98)
    distributions.BasePointTransformation(end_points)
    source = sources.AperatureSource(
        3, start_points, end_points, [drawing.YELLOW], dense=False,
        extra_fields={"object_coords": ("start_point", start_points, "points")})

    def surface(k, flip, sign):
        zp = mt.hexagonal_mesh(1.0, k)
        zp.rotate_y(90)
        zp.rotate_x(90)
        r2 = zp.points[:, 1] ** 2 + zp.points[:, 2] ** 2
        return boundaries.ParametricTriangleBoundary(
            zp, boundaries.FromVectorVG((1, 0, 0)), flip_norm=flip,
            initial_parameters=sign * (0.1 + 0.15 * (1 - r2)),
            material_dict={"mat_in": 1, "mat_out": 0})

    front = surface(k_front, True, -1.0)
    back = surface(k_back, False, +1.0)
    target = boundaries.ManualTriangleBoundary(
        mesh=mt.plane(center=(10, 0, 0), direction=(1, 0, 0), i_size=100, j_size=100))
    target.frozen = True
    system = engine.OpticalSystem3D()
    system.optical = [front, back]
    system.targets = [target]
    system.sources = [source]
    system.materials = [{"n": materials.vacuum}, {"n": materials.acrylic}]
    system.update()
    eng = engine.OpticalEngine(
        3, [operation.StandardReaction()], compile_active_rays=False,
        simple_ray_inheritance={"wavelength", "object_coords"}, ray_dtype=ray_dtype,
        accelerate=accelerate)
    eng.optical_system = system
    eng.validate_system()
    return eng, system, [front.parameters, back.parameters]


def error_function(engine):
    """Image-forming error of dev/hexalens.py:154-168 (inner goal, magnification 1)."""
    fin = engine.finished_rays
    out = torch.stack([fin["y_end"], fin["z_end"]], dim=1).double()
    goal = -fin["object_coords"][:, 1:]
    return (out - goal) ** 2


def cpu_baseline(seconds_budget=20.0):
    """The reference algorithm restated (oracle: dense (M,N) float64 torch ops, forward +
    autograd) on the host cores, on a bounded sample of the same scene."""
    import scene_util
    import oracle_util
    from oracle import tracer
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a one-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    n = 256
    scene = scene_util.lens_scene(n, k_front=41, k_back=9)
    t0 = time.time()
    tests = 0
    reps = 0
    while True:
        system, (p_f, p_b), _ = oracle_util.lens_oracle(scene)
        ref = tracer.ray_trace(system, oracle_util.source_dict(scene["rays"], scene["wavelength"]),
                               max_iterations=3, inherit=("wavelength", "ray_id"), chunk=256)
        fin = ref["finished"]
        goal = torch.tensor(scene["goal"], dtype=torch.float64)[fin["ray_id"].long()]
        err = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
        torch.autograd.grad(err, [p_f, p_b])
        m = system.merged["xp"].shape[0]
        n_act = n + sum(int(h["x_start"].shape[0]) for h in [ref["active"]] if h)
        tests += n_act * m
        reps += 1
        if time.time() - t0 > seconds_budget:
            break
    dt = time.time() - t0
    return {
        "value": tests / dt, "unit": "tests/s", "cores": cores, "kind": "port",
        "sample": f"{reps} x (fwd + autograd) of the oracle (torch-CPU float64, dense (M,N) "
                  f"temporaries, 256-ray chunks) on {n} rays x 10574 faces, 3 passes; "
                  f"{dt:.1f} s wall",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1_000_000,
                    help="rays per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU traces --rays rays of a global source of rays x N; "
                         "strong: --rays rays are split over the GPUs")
    ap.add_argument("--k-front", type=int, default=41)
    ap.add_argument("--k-back", type=int, default=9)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--trace-mode", choices=["auto", "all-pairs", "group", "sort"], default="auto",
                    help="how ray-face pairs are culled before the exact float64 test (results "
                         "are identical in every mode): all-pairs = float32 bounding-sphere "
                         "filter on every pair; group = two/three-level sphere hierarchy over "
                         "k-d face clusters; sort = clusters + Morton-sorted rays; auto = the "
                         "engine's default (group)")
    ap.add_argument("--accelerate", action="store_true", help="alias of --trace-mode sort")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the separately reported legs that time the other trace modes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tfrt hot path has no CPU fallback")

    import tensorflowraytrace_amd as tfa
    from tensorflowraytrace_amd import _lib, distributed as tdist
    import tfrt.optimizer as optimizer

    rank, world, local = tdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device((local % torch.cuda.device_count()) if world > 1 else 0)
    tfa.set_device(f"cuda:{torch.cuda.current_device()}")
    ray_dtype = torch.float32 if args.dtype == "f32" else torch.float64

    if args.accelerate:
        args.trace_mode = "sort"
    lib = _lib.lib()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_leg(global_rays, trace_mode, profile):
        """Build the scene with `global_rays` source rays (each rank traces its contiguous
        shard), run warmup + exactly `steps` optimiser steps between barriers and return
        (max-over-ranks seconds, tests summed over ranks, engine, faces, mode, launch ms)."""
        eng, system, params = build_scene(global_rays, args.k_front, args.k_back, ray_dtype,
                                          accelerate=trace_mode)
        opt = optimizer.SGD_Optimizer(eng, params, error_function, trace_depth=3,
                                      learning_rate=1e-6, grad_clip=1e-3)
        opt.suppress_warnings = True
        for _ in range(args.warmup):
            opt.single_step(None)
        barrier()
        if profile:
            lib.tfrt_profile_enable(1)
        tests_local = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            opt.single_step(None)
            tests_local += eng.last_trace["n_tests"]
        barrier()
        dt = time.perf_counter() - t0
        kernel_ms = []
        if profile:
            import ctypes
            buf = (ctypes.c_float * 4096)()
            nrec = lib.tfrt_profile_read(buf, 4096)
            lib.tfrt_profile_enable(0)
            kernel_ms = [buf[i] for i in range(max(nrec, 0))]
        tests_total = float(tests_local)
        if world > 1:
            stats = torch.tensor([dt, float(tests_local)], dtype=torch.float64, device="cuda")
            tmax = stats[:1].clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            tsum = stats[1:].clone()
            torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
            dt, tests_total = float(tmax.item()), float(tsum.item())
        return dt, tests_total, eng, int(system._merged_face_verts.shape[0]), \
            eng._trace_mode(system), kernel_ms

    # weak scaling (default): every GPU traces `--rays` rays, the global source has rays x N;
    # strong: the same `--rays` rays are split over the GPUs
    global_rays = args.rays * world if args.scaling == "weak" else args.rays
    dt, tests_total, eng, M, mode, kernel_ms = timed_leg(global_rays, args.trace_mode, True)
    counts = eng.last_trace["counts"]
    del eng
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_extra_legs:
        # separately reported (never `value`): the N=1 workload itself split over the ranks
        torch.cuda.empty_cache()
        dts, tests_s, eng_s, _, _, _ = timed_leg(args.rays, args.trace_mode, False)
        del eng_s
        strong = {"global_rays": args.rays, "ms_per_step": dts / args.steps * 1e3,
                  "tests_per_s": tests_s / dts,
                  "note": "the N=1 workload split over the ranks; the step is host-bound below "
                          "~250k rays per rank (DESIGN.md section 6)"}

    if rank != 0:
        return

    n_active = [int(c[:4].sum()) for c in counts]          # rays entering each pass (rank 0)
    launches = len(kernel_ms)
    avg_ms = float(np.mean(kernel_ms)) if launches else float("nan")
    tests_per_launch = float(np.mean([n * M for n in n_active])) if n_active else 0.0
    alg_flops = tests_per_launch * FLOPS_PER_TEST
    alg_bytes = float(np.mean(n_active)) * BYTES_PER_RAY_FWD + M * BYTES_PER_FACE
    achieved_tf = alg_flops / (avg_ms * 1e-3) / 1e12 if launches else float("nan")
    kernel_name = {"all-pairs": "tfrt::k_intersect3d", "group": "tfrt::k_intersect_group",
                   "sort": "tfrt::k_intersect_cull"}[mode]
    executed = {"all-pairs": 8.75, "group": 0.71, "sort": None}[mode]
    roofline = {
        "kernel": kernel_name,
        "bound": "valu",
        "achieved": achieved_tf, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved_tf / PEAK_VALU_TFLOPS,
        "avg_launch_ms": avg_ms, "launches_timed": launches,
        "tests_per_launch": tests_per_launch,
        "model": "SURVEY.md 8d: 45 flop per ray-face pair (Moeller-Trumbore) x pairs decided per "
                 "launch / launch time.  The kernel decides pairs through conservative float32 "
                 "bounding-sphere tests (a hierarchy over face clusters in the default mode) and "
                 "runs the exact float64 test on the survivors only, so this algorithmic rate "
                 "exceeds the VALU peak; executed_valu_ops_per_test and valu_issue_utilisation "
                 "(PMC, profiles/) describe the executed work",
        "executed_valu_ops_per_test": executed,
        "valu_issue_utilisation": {"all-pairs": 0.84, "group": 0.54, "sort": None}[mode],
        "hbm_achieved_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else float("nan"),
        "hbm_frac": (alg_bytes / (avg_ms * 1e-3) / 1e9) / PEAK_HBM_GBPS if launches else float("nan"),
        "traffic": None,
    }
    # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
    # of the same kernel on the same workload, committed under profiles/
    tpath = os.path.join(ROOT, "profiles", {"all-pairs": "r01_traffic.json",
                                             "group": "r01_traffic_group.json"}.get(mode, "none"))
    if (os.path.exists(tpath) and args.rays == 1_000_000 and args.gpus == 1
            and args.k_front == 41):
        tj = json.load(open(tpath))
        roofline["traffic"] = (tj["FETCH_SIZE_KB"] + tj["WRITE_SIZE_KB"]) * 1024.0
        roofline["traffic_unit"] = "bytes per launch (PMC FETCH_SIZE + WRITE_SIZE, uncorrected)"
        roofline["traffic_source"] = os.path.relpath(tpath, ROOT)
    line = {
        "metric": "ray-surface intersection tests/sec (fwd+bwd)",
        "value": tests_total / dt,
        "unit": "tests/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": args.dtype + " ray state, f32 filter + f64 decisions",
        "data": "synthetic",
        "config": {
            "workload": "cfg4: 1M-ray aperture source x 2-surface parametric hex lens "
                        "(10086+486 faces) + 2-face target, SGD_Optimizer.single_step, "
                        "trace_depth 3",
            "rays_per_gpu": global_rays // world, "global_rays": global_rays, "faces": M,
            "trace_depth": 3,
            "trace_mode": {"all-pairs": "all-pairs float32 sphere filter",
                           "group": "sphere hierarchy over k-d face clusters (default)",
                           "sort": "face clusters + Morton-sorted rays"}[mode],
            "parallelism": f"rays sharded over {args.gpus} GPU(s), 1 RCCL all-reduce/step",
        },
        "roofline": roofline,
    }
    if strong is not None:
        line["strong_scaling"] = strong
    if world == 1 and not args.no_extra_legs:
        # separately reported legs (never `value`): the same step in the other trace modes.
        # Results are bit-identical in every mode; pairs = N_active x M as in `value`.
        legs = {}
        for other in ("all-pairs", "group", "sort"):
            if other == mode:
                continue
            torch.cuda.empty_cache()
            eng2, system2, params2 = build_scene(args.rays, args.k_front, args.k_back, ray_dtype,
                                                 accelerate=other)
            opt2 = optimizer.SGD_Optimizer(eng2, params2, error_function, trace_depth=3,
                                           learning_rate=1e-6, grad_clip=1e-3)
            opt2.suppress_warnings = True
            for _ in range(args.warmup):
                opt2.single_step(None)
            torch.cuda.synchronize()
            pairs = 0
            t1 = time.perf_counter()
            for _ in range(args.steps):
                opt2.single_step(None)
                pairs += eng2.last_trace["n_tests"]
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            legs[other] = {"ms_per_step": dt2 / args.steps * 1e3, "tests_per_s": pairs / dt2}
            del eng2, system2, params2, opt2
        line["other_trace_modes"] = legs
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
