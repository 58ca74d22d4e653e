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
all-reduce.  Default is STRONG scaling (BASELINE configs[3]: the same 1M rays split over the
ranks; value = tests of all ranks / max-over-ranks time); `--scaling weak` gives every GPU 1M rays
and is also reported as a side field.

The step runs as the fused launch sequence of tensorflowraytrace_amd/fused_step.py (error function
stated as a GoalError, captured in a HIP graph); `--step-mode generic` times the same step with the
error function as arbitrary torch code instead.

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
VALU_SIMDS = 1024          # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
CYCLES_PER_WAVE_OP = 2.0   # MI355X_MICROARCH.md: v_fma_f32 (wave64) throughput, SIMD-32
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E spec peak (6290 measured copy)
FLOPS_PER_TEST = 45        # SURVEY.md section 8d algorithmic cost model (Moeller-Trumbore)
BYTES_PER_RAY_FWD = 64     # 32 B read + 32 B written per active ray per pass
BYTES_PER_RAY_BWD = 116    # tape 32 + upstream 24 + downstream 24 + 9 atomics 36
BYTES_PER_FACE = 48


def build_scene(n_rays, k_front, k_back, ray_dtype, accelerate="auto", random_rays=False,
                coherent="auto"):
    import tensorflowraytrace_amd as tfa
    import tfrt.boundaries as boundaries
    import tfrt.distributions as distributions
    import tfrt.drawing as drawing
    import tfrt.engine as engine
    import tfrt.materials as materials
    import tfrt.mesh_tools as mt
    import tfrt.operation as operation
    import tfrt.sources as sources

    # random_rays: the source of dev/hexalens.py:36-48 -- RandomUniformCircle distributions, re-drawn
    # by every optical_system.update(), i.e. every optimiser step
    circle = distributions.RandomUniformCircle if random_rays else distributions.StaticUniformCircle
    start_points = circle(n_rays, 0.2)
    distributions.BasePointTransformation(start_points, translation=(-10, 0, 0))
    end_points = circle(n_rays, 0.98)
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
        accelerate=accelerate, coherent=coherent)
    eng.optical_system = system
    eng.validate_system()
    return eng, system, [front.parameters, back.parameters]


def goal(src):
    """Image-forming goal of dev/hexalens.py:154-157 (inner goal, magnification 1): the image
    point of a ray's object point, one row per source ray."""
    return -src["object_coords"][:, 1:]


def make_error_function():
    """error = squared_difference(stack(finished y_end, z_end), goal) (dev/hexalens.py:144-168),
    stated as a GoalError so the optimiser can run the step as one fixed launch sequence; it is
    also an ordinary error_function(engine) (``--generic-step`` times that path).  The goal of a
    ray is a function of that ray's own fields (rowwise)."""
    import tfrt.optimizer as optimizer
    return optimizer.GoalError(("y_end", "z_end"), goal, rowwise=True)


def cpu_baseline(seconds_budget=20.0):
    """The reference algorithm restated (oracle: dense (M,N) float64 torch ops, forward +
    autograd) on the host cores, on a bounded sample of the same scene: consecutive 256-ray chunks
    of the source (each chunk a forward + backward of its own: the gradient is a sum over rays,
    and autograd keeps ~100 dense (10574 x chunk) float64 temporaries alive per chunk)."""
    import scene_util
    import oracle_util
    from oracle import tracer
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a one-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    # 256-ray chunks: a dense (10574 x 256) float64 temporary is 21.6 MB, below glibc's 32 MB mmap
    # threshold ceiling, so freed temporaries are reused instead of being unmapped and
    # page-faulted in again on every one of the ~125 dense ops (4-20x slower at 512 rays)
    chunk, warm_chunks = 256, 2
    scene = scene_util.lens_scene(65_536, k_front=41, k_back=9)
    tests = 0
    rays_done = 0
    t0 = None
    k = 0
    while (k + 1) * chunk <= scene["rays"].shape[1]:
        if k == warm_chunks:
            t0 = time.time()          # the first chunks pay first-touch page faults: not timed
        lo, hi = k * chunk, (k + 1) * chunk
        k += 1
        system, (p_f, p_b), _ = oracle_util.lens_oracle(scene)
        ref = tracer.ray_trace(
            system, oracle_util.source_dict(scene["rays"][:, lo:hi], scene["wavelength"][lo:hi]),
            max_iterations=3, inherit=("wavelength", "ray_id"), chunk=chunk)
        fin = ref["finished"]
        goal = torch.tensor(scene["goal"][lo:hi], dtype=torch.float64)[fin["ray_id"].long()]
        err = ((fin["y_end"] - goal[:, 0]) ** 2 + (fin["z_end"] - goal[:, 1]) ** 2).sum()
        torch.autograd.grad(err, [p_f, p_b])
        if t0 is None:
            continue
        m = system.merged["xp"].shape[0]
        n_act = chunk + (int(ref["active"]["x_start"].shape[0]) if ref["active"] else 0)
        tests += n_act * m
        rays_done += chunk
        if time.time() - t0 > seconds_budget:
            break
    dt = time.time() - t0
    return {
        "value": tests / dt, "unit": "tests/s", "cores": cores, "kind": "port",
        "sample": f"{rays_done} consecutive source rays (of a 65,536-ray instance of the same "
                  f"scene) x 10574 faces, 3 passes, forward + autograd of the oracle (torch-CPU "
                  f"float64, dense (M,N) temporaries) in {chunk}-ray chunks after {warm_chunks} "
                  f"untimed warm-up chunks; {dt:.1f} s wall",
    }


def _source_hash():
    """Hash of the kernel sources: profiles/*.json record it, so PMC-derived numbers are only
    reported for the kernels they were collected on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tensorflowraytrace_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: start N fresh worker
    processes (one rank per GPU) BEFORE this process touches a GPU, relay their output (rank 0
    prints the JSON line) and return the launcher's exit status.  Workers never re-exec."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


MIN_TIMED_SECONDS = 0.2


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1_000_000,
                    help="rays in total (strong scaling) or per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong: --rays rays are split over the GPUs (BASELINE configs[3]); "
                         "weak: every GPU traces --rays rays of a global source of rays x N")
    ap.add_argument("--k-front", type=int, default=41)
    ap.add_argument("--k-back", type=int, default=9)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--step-mode", choices=["graph", "fused", "generic"], default="graph",
                    help="graph: fused launch sequence replayed from a HIP graph (default); fused: "
                         "the same sequence launched eagerly; generic: error function as arbitrary "
                         "torch code through autograd (reads the ray counts back every step)")
    ap.add_argument("--trace-mode", choices=["auto", "all-pairs", "group"], default="auto",
                    help="how ray-face pairs are culled before the exact float64 test (results "
                         "are identical in every mode): all-pairs = float32 bounding-sphere "
                         "filter on every pair; group = sphere hierarchy over k-d face "
                         "clusters; auto = the engine's default (group)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5a", "cfg5b"], default="cfg4",
                    help="BASELINE.json configuration: cfg4 (default) is the headline; the others "
                         "emit a line of the same shape for their workload (bench_configs.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the separately reported legs (all-pairs mode, float64 state, ...)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--collective-in-graph", action="store_true",
                    help="several ranks: capture the RCCL all-reduce inside the step's graph "
                         "(FusedStep.capture_collective; default: called between two graphs -- the "
                         "captured form has only been replayed with a one-rank group so far)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: one process cannot stand for N ranks
        raise SystemExit(spawn_ranks(args.gpus, argv))

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tfrt hot path has no CPU fallback")
    if args.config != "cfg4":
        if args.gpus != 1:
            raise SystemExit("--config other than cfg4 is a single-GPU measurement")
        import bench_configs
        bench_configs.run(args.config, args)
        return

    import tensorflowraytrace_amd as tfa
    from tensorflowraytrace_amd import _lib, distributed as tdist
    import tfrt.optimizer as optimizer

    if args.collective_in_graph:
        from tensorflowraytrace_amd import fused_step
        fused_step.FusedStep.capture_collective = True
    rank, world, local = tdist.init_from_env()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {world} rank(s) are running (WORLD_SIZE)")
    backend = torch.distributed.get_backend() if tdist.is_distributed() else None
    torch.cuda.set_device((local % torch.cuda.device_count()) if world > 1 else 0)
    tfa.set_device(f"cuda:{torch.cuda.current_device()}")
    lib = _lib.lib()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_leg(global_rays, trace_mode, step_mode, dtype, profile=False, random_rays=False,
                  coherent="auto"):
        """Build the scene with `global_rays` source rays (each rank traces its contiguous
        shard), run warmup + exactly `steps` optimiser steps between barriers.  Returns a dict:
        seconds (max over ranks), tests (sum over ranks), per-launch ms of the dominant kernel
        (only with `profile`, which needs eagerly launched kernels), pass counts, faces, mode."""
        ray_dtype = torch.float32 if dtype == "f32" else torch.float64
        eng, system, params = build_scene(global_rays, args.k_front, args.k_back, ray_dtype,
                                          accelerate=trace_mode, random_rays=random_rays,
                                          coherent=coherent)
        opt = optimizer.SGD_Optimizer(
            eng, params, make_error_function(), trace_depth=3, learning_rate=1e-6, grad_clip=1e-3,
            fused=False if step_mode == "generic" else "auto",
            graph="auto" if step_mode == "graph" else False,
            speculative=step_mode == "generic")   # (the bench's error function is pure)
        opt.suppress_warnings = True
        for _ in range(args.warmup):
            opt.single_step(None)
        fs = opt._fused_step
        barrier()
        # at least MIN_TIMED_SECONDS inside the timed region: the requested step count is raised
        # when K steps would be shorter (every rank takes the same count: max over ranks)
        t0 = time.perf_counter()
        for _ in range(3):
            opt.single_step(None)
        torch.cuda.synchronize()
        est = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device="cuda")
        if world > 1:
            torch.distributed.all_reduce(est, op=torch.distributed.ReduceOp.MIN)
        steps = max(args.steps, int(np.ceil(MIN_TIMED_SECONDS / max(float(est.item()), 1e-6))))
        barrier()
        if profile:
            lib.tfrt_profile_enable(1)
        tests0 = int(fs.tests_total.item()) if fs is not None else 0
        tests_local = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            opt.single_step(None)
            if fs is None:
                tests_local += eng.last_trace["n_tests"]
        barrier()
        dt = time.perf_counter() - t0
        if fs is not None:
            tests_local = int(fs.tests_total.item()) - tests0
        kernel_ms = []
        if profile:
            import ctypes
            buf = (ctypes.c_float * 8192)()
            kernel_ms = {}
            for kind, name in enumerate(("intersect", "react", "backward", "accumulate")):
                nrec = lib.tfrt_profile_read_kind(kind, buf, 8192)
                kernel_ms[name] = [buf[i] for i in range(max(nrec, 0))]
            lib.tfrt_profile_enable(0)
        tests_total = float(tests_local)
        if world > 1:
            stats = torch.tensor([dt, float(tests_local)], dtype=torch.float64, device="cuda")
            tmax = stats[:1].clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            tsum = stats[1:].clone()
            torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
            dt, tests_total = float(tmax.item()), float(tsum.item())
        out = dict(dt=dt, steps=steps, tests=tests_total, kernel_ms=kernel_ms,
                   counts=eng.last_trace["counts"], M=int(system._merged_face_verts.shape[0]),
                   mode=eng._trace_mode(system),
                   visiting=getattr(eng, "_order_cache", None) is not None,
                   graph_replays=fs.graph_replays if fs is not None else 0,
                   collective_in_graph=bool(fs.collective_in_graph) if fs is not None else None,
                   capture_error=repr(fs.capture_error) if fs is not None and fs.capture_error
                   else None)
        del eng, system, params, opt
        torch.cuda.empty_cache()
        return out

    per_rank = args.scaling == "weak"
    global_rays = args.rays * world if per_rank else args.rays
    main_leg = timed_leg(global_rays, args.trace_mode, args.step_mode, args.dtype)
    side = {}
    if world > 1 and not args.no_extra_legs:
        # the other scaling convention, separately reported (never `value`)
        other_rays = args.rays if per_rank else args.rays * world
        leg = timed_leg(other_rays, args.trace_mode, args.step_mode, args.dtype)
        side["weak_scaling" if not per_rank else "strong_scaling"] = {
            "global_rays": other_rays, "ms_per_step": leg["dt"] / leg["steps"] * 1e3,
            "tests_per_s": leg["tests"] / leg["dt"]}
    # per-launch time of the dominant kernel, HIP events on the launch stream: needs eagerly
    # launched kernels (events cannot sit inside a replayed graph), so the same step is run
    # `steps` more times as the eager fused sequence -- same kernels, same launches, same data
    prof_leg = None
    if world == 1:
        prof_leg = timed_leg(global_rays, args.trace_mode,
                             "generic" if args.step_mode == "generic" else "fused", args.dtype,
                             profile=True)

    if rank != 0:
        return

    M, mode, counts = main_leg["M"], main_leg["mode"], main_leg["counts"]
    n_active = [int(c[:4].sum()) for c in counts]          # rays entering each pass (rank 0)
    P = len(n_active)
    visiting = bool(main_leg["visiting"]) and mode == "group"
    kernel_name = ("tfrt::k_intersect_beam" if visiting else
                   {"all-pairs": "tfrt::k_intersect3d", "group": "tfrt::k_intersect_group"}[mode])
    roofline = {"kernel": kernel_name, "bound": "valu"}
    same_workload = (args.rays == 1_000_000 and world == 1 and args.k_front == 41
                     and args.k_back == 9 and args.dtype == "f32")
    src_hash = _source_hash()

    def pmc_of(tag):
        """Counters of one kernel from profiles/ (separate rocprofv3 --pmc runs of the same step,
        scratch/collect_profiles.sh) -- only for the kernel sources and workload they were
        collected on."""
        path = os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % tag)
        if not (same_workload and os.path.exists(path)):
            return None, path
        doc = json.load(open(path))
        return (doc if doc.get("source_hash") == src_hash else None), path

    def traffic_of(doc):
        fetch = float(np.mean([p["FETCH_SIZE_KB"] for p in doc["passes"]])) * 1024.0
        write = float(np.mean([p["WRITE_SIZE_KB"] for p in doc["passes"]])) * 1024.0
        return 2.0 * fetch + write      # guide: FETCH_SIZE counts half of wide reads on gfx950

    kms = prof_leg["kernel_ms"] if prof_leg is not None else {}
    if kms.get("intersect"):
        ms = np.asarray(kms["intersect"], dtype=np.float64)
        ms = ms[:len(ms) // P * P].reshape(-1, P)
        per_pass_ms = ms.mean(axis=0)
        avg_ms = float(ms.mean())
        tests_per_launch = float(np.mean([n * M for n in n_active]))
        alg_bytes = float(np.mean(n_active)) * BYTES_PER_RAY_FWD + M * BYTES_PER_FACE
        roofline.update({
            "avg_launch_ms": avg_ms, "per_pass_launch_ms": [float(x) for x in per_pass_ms],
            "launches_timed": int(ms.size),
            "timed_by": "HIP events on the launch stream (tfrt_profile_*), eager fused sequence "
                        "run after the timed region with the same steps"
                        + ("; one record = the pass's k_intersect_beam launch + the "
                           "k_intersect_group launch for left-over wavefronts (none on this "
                           "workload: it reads one counter and retires)" if visiting else ""),
            "tests_per_launch": tests_per_launch,
            # SURVEY.md 8d cost model, kept as a separately named field: pairs DECIDED x 45 flop.
            # The kernel decides pairs through conservative sphere tests, so this is not a
            # utilisation of anything (it exceeds the VALU peak)
            "algorithmic_tflops": tests_per_launch * FLOPS_PER_TEST / (avg_ms * 1e-3) / 1e12,
            "hbm": {
                "algorithmic_bytes_per_launch": alg_bytes,
                "achieved_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9,
                "peak_GBps": PEAK_HBM_GBPS,
                "frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            },
        })
        # executed work: VALU wave-instructions per launch from the PMC passes, combined with the
        # launch time measured live above
        tag = "beam" if visiting else mode.replace("-", "_")
        pmc, pmc_path = pmc_of(tag)
        if pmc is not None:
            valu = float(np.mean([p["SQ_INSTS_VALU"] for p in pmc["passes"]]))
            issue_rate = valu / (avg_ms * 1e-3)                     # wave-instructions / s
            peak = VALU_SIMDS * CLOCK_HZ / CYCLES_PER_WAVE_OP
            traffic = traffic_of(pmc)
            roofline.update({
                "achieved": issue_rate, "peak": peak, "unit": "VALU wave-instructions/s",
                "frac": issue_rate / peak,
                "peak_model": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 op "
                              "(MI355X_MICROARCH.md, v_fma_f32 row)",
                "valu_wave_instructions_per_launch": valu,
                "salu_per_valu": float(np.mean([p["SQ_INSTS_SALU"] / p["SQ_INSTS_VALU"]
                                                for p in pmc["passes"]])),
                "lds_bank_conflict_ratio": float(np.mean(
                    [p["SQ_LDS_BANK_CONFLICT"] / max(p["SQ_LDS_IDX_ACTIVE"], 1.0)
                     for p in pmc["passes"]])),
                "executed_valu_lane_ops_per_test": valu * 64.0 / tests_per_launch,
                "wait_fraction_of_wave_cycles": float(np.mean(
                    [p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"] for p in pmc["passes"]])),
                "traffic": traffic,
                "traffic_unit": "HBM-side bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE "
                                "(gfx950 correction of the guide for wide reads)",
                "pmc_source": os.path.relpath(pmc_path, ROOT),
            })
            roofline["hbm"]["traffic_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS
        else:
            roofline.update({
                "achieved": None, "peak": None, "unit": "VALU wave-instructions/s", "frac": None,
                "traffic": None,
                "pmc_note": "no PMC profile of this kernel source / workload under profiles/ "
                            "(regenerate with scratch/collect_profiles.sh); counter-derived "
                            "fields withheld"})
        # every hot kernel of the step: live launch time, the roof that bounds it, SURVEY 8d's
        # algorithmic bytes beside the counter traffic
        n_fwd = float(np.mean(n_active))
        # coherent rays: the reverse sweep is ONE launch per step (as many records as react has / P)
        bwd_one = len(kms.get("backward") or []) * P <= len(kms.get("react") or []) + P - 1
        rows = [dict(name=kernel_name, kind="intersect", launches_per_step=P, bound="valu",
                     avg_ms=avg_ms, frac=roofline.get("frac"),
                     algorithmic_bytes=alg_bytes, traffic=roofline.get("traffic"))]
        for kind, kname, per_step, alg, note in (
                ("react", "tfrt::k_react3d", P, n_fwd * BYTES_PER_RAY_FWD,
                 "SURVEY 8d forward bytes: 64 B per ray entering the pass"),
                ("backward", "tfrt::k_backward_chain" if bwd_one else "tfrt::k_backward3d",
                 1 if bwd_one else P,
                 (float(sum(n_active)) if bwd_one else n_fwd) * BYTES_PER_RAY_BWD,
                 "SURVEY 8d backward bytes: 116 B per ray entering a pass" +
                 (", all passes in the one launch (coherent rays: a lane walks its ray's chain of "
                  "slots back with the child gradient in registers; the face-gradient terms are summed "
                  "per wavefront in LDS and leave as one atomic per face and term)" if bwd_one else
                  " (one launch per pass)")),
                ("accumulate", "tfrt::k_face_accumulate", 1, float(sum(n_active)) * 40.0,
                 "the stash read once: 36 B of terms + 4 B face index per ray and pass")):
            v = np.asarray(kms.get(kind) or [], dtype=np.float64)
            if not v.size:      # (coherent rays: no stash, no accumulate launch)
                continue
            k_ms = float(v.mean())
            doc, path = pmc_of(kind)
            traffic = traffic_of(doc) if doc is not None else None
            rows.append(dict(
                name=kname, kind=kind, launches_per_step=per_step, bound="hbm", avg_ms=k_ms,
                algorithmic_bytes=alg, algorithmic_note=note,
                achieved_GBps=alg / (k_ms * 1e-3) / 1e9,
                frac=alg / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                traffic=traffic,
                traffic_frac=(traffic / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS)
                if traffic is not None else None,
                pmc_source=os.path.relpath(path, ROOT) if doc is not None else None))
        roofline["kernels"] = rows
        roofline["kernels_note"] = ("avg_ms: HIP events per launch (eager fused sequence); "
                                    "frac: VALU issue fraction (intersect) or algorithmic bytes / "
                                    "time / 8 TB/s (HBM-bound kernels); traffic: 2 x FETCH_SIZE + "
                                    "WRITE_SIZE per launch from profiles/ when collected on these "
                                    "kernel sources, else null")
        step_ms = sum(r["avg_ms"] * r["launches_per_step"] for r in rows)
        roofline["hot_kernels_ms_per_step"] = step_ms
    dt, tests_total = main_leg["dt"], main_leg["tests"]
    line = {
        "metric": "ray-surface intersection tests/sec (fwd+bwd)",
        "value": tests_total / dt,
        "unit": "tests/s",
        "n_gpus": world,
        "steps": main_leg["steps"],
        "steps_requested": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / main_leg["steps"] * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": args.dtype + " ray state, f32 filter + f64 decisions",
        "dtype_note": "every hit/miss and nearest-face decision, hit point, Snell step and the reverse "
                      "sweep's arithmetic and sums are float64; the ray state between passes and the "
                      "reverse sweep's per-ray intermediates are STORED in " + args.dtype +
                      " (other_legs.f64_ray_state: everything float64)",
        "data": "synthetic",
        "config": {
            "workload": "cfg4: 1M-ray aperture source x 2-surface parametric hex lens "
                        "(10086+486 faces) + 2-face target, SGD_Optimizer.single_step, "
                        "trace_depth 3",
            "rays_per_gpu": global_rays // world, "global_rays": global_rays, "faces": M,
            "trace_depth": 3,
            "trace_mode": {"all-pairs": "all-pairs float32 sphere filter",
                           "group": "sphere hierarchy over k-d face clusters (default)"}[mode]
            + (", rays visited in a coherent (Hilbert) order: wavefronts share one walk "
               "(k_intersect_beam)" if visiting else ""),
            "step_mode": {"graph": "fused launch sequence (GoalError), HIP-graph replay",
                          "fused": "fused launch sequence (GoalError), eager launches",
                          "generic": "error function as torch code through autograd"}[
                              args.step_mode],
            "graph_replays": main_leg["graph_replays"],
            "ranks": world, "backend": backend,
            "collective_in_graph": main_leg.get("collective_in_graph"),
            "parallelism": (f"rays sharded over {world} ranks (one per GPU), 1 {backend} "
                            f"all-reduce of the parameter gradients per step")
            if tdist.is_distributed() else "single GPU, no collective",
        },
        "value_note": "tests = ray-face pairs DECIDED (sum over passes of N_active x M, counted "
                      "by the kernels); the hierarchy decides most pairs without executing a "
                      "per-pair test -- ms_per_step is the comparable number, and "
                      "other_legs.all_pairs is the same step with every pair executed",
        "roofline": roofline,
    }
    # whole step against the HBM roof: SURVEY 8d bytes (64 B forward + 116 B backward per ray
    # entering a pass, 48 B per face and pass) / ms_per_step / 8 TB/s
    step_bytes = (sum(n_active) * (BYTES_PER_RAY_FWD + BYTES_PER_RAY_BWD) * world   # (rank 0's shard x ranks)
                  + P * M * BYTES_PER_FACE * world)
    line["roofline"]["hbm_frac_step"] = step_bytes / (dt / main_leg["steps"]) / 1e9 / PEAK_HBM_GBPS
    line["roofline"]["hbm_frac_step_note"] = (
        "SURVEY 8d algorithmic bytes of one optimiser step (%.1f MB) / ms_per_step / 8 TB/s"
        % (step_bytes / 1e6))
    if main_leg["capture_error"]:
        line["config"]["graph_capture_error"] = main_leg["capture_error"]
    line.update(side)
    if world == 1 and not args.no_extra_legs:
        # separately reported legs (never `value`); results are identical in every trace mode
        legs = {}
        if mode != "all-pairs":
            leg = timed_leg(args.rays, "all-pairs", args.step_mode, args.dtype)
            legs["all_pairs"] = {"ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                                 "tests_per_s": leg["tests"] / leg["dt"],
                                 "note": "every ray-face pair goes through the float32 sphere test"}
        other_dt = "f64" if args.dtype == "f32" else "f32"
        leg = timed_leg(args.rays, args.trace_mode, args.step_mode, other_dt)
        legs[other_dt + "_ray_state"] = {"ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                                         "tests_per_s": leg["tests"] / leg["dt"]}
        if args.step_mode != "generic":
            leg = timed_leg(args.rays, args.trace_mode, "generic", args.dtype)
            legs["generic_step"] = {"ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                                    "tests_per_s": leg["tests"] / leg["dt"],
                                    "note": "same error function as arbitrary torch code (rays "
                                            "ordered on the device, every ray set handed back in "
                                            "the reference's order)"}
            # the reference's own optimisation workload: the source is re-drawn at every step
            # (dev/hexalens.py:36-48 RandomUniformCircle, optimizer.py:217 update() per step)
            leg = timed_leg(args.rays, args.trace_mode, args.step_mode, args.dtype,
                            random_rays=True)
            legs["random_source"] = {
                "ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                "tests_per_s": leg["tests"] / leg["dt"],
                "graph_replays": leg["graph_replays"], "capture_error": leg["capture_error"],
                "ordered": bool(leg["visiting"]),
                "note": "source rays drawn in place every step (tfrt_source3d_generate), ordered "
                        "every step (tfrt_ray_order), same launch graph"}
            leg = timed_leg(args.rays, args.trace_mode, args.step_mode, args.dtype,
                            coherent=False)
            legs["natural_order"] = {"ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                                     "tests_per_s": leg["tests"] / leg["dt"],
                                     "note": "the headline step with the rays in source order "
                                             "(k_intersect_group, stash + k_face_accumulate)"}
        line["other_legs"] = legs
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
