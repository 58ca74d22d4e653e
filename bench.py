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


def make_rowwise_error_function():
    """The same error as arbitrary torch code, stated row by row (optimizer.RowwiseError): the
    optimiser evaluates it on fixed-shape tensors inside the step's launch graph."""
    import tfrt.optimizer as optimizer

    def fn(rays):
        out = torch.stack([rays["y_end"], rays["z_end"]], dim=1).double()
        return (out + rays["object_coords"][:, 1:]) ** 2
    return optimizer.RowwiseError(fn)


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

    def timed_leg(*a, **kw):
        # (several ranks: a leg that does not finish -- a collective captured on one rank and not
        # on another hangs, it does not raise -- ends this process with status 124 and the
        # launcher with it, instead of blocking the node)
        with tdist.Watchdog(float(os.environ.get("TFRT_BENCH_LEG_TIMEOUT", 300)),
                            "a bench leg (warm-up, graph capture, timed steps)"):
            return _timed_leg(*a, **kw)

    def _timed_leg(global_rays, trace_mode, step_mode, dtype, profile=False, random_rays=False,
                  coherent="auto", exact_steps=False):
        """Build the scene with `global_rays` source rays (each rank traces its contiguous
        shard), run warmup steps, then time optimiser steps between barriers: EXACTLY --steps of
        them with `exact_steps` (the headline leg: the contract's K), else at least
        MIN_TIMED_SECONDS worth (side legs).  The headline leg then runs a second, longer loop
        (`steady`: >= MIN_TIMED_SECONDS, reported beside the line, never `value`).  Returns a dict:
        seconds (max over ranks), tests (sum over ranks), per-launch ms of the hot kernels (only
        with `profile`, which needs eagerly launched kernels), pass counts, faces, mode."""
        ray_dtype = torch.float32 if dtype == "f32" else torch.float64
        eng, system, params = build_scene(global_rays, args.k_front, args.k_back, ray_dtype,
                                          accelerate=trace_mode, random_rays=random_rays,
                                          coherent=coherent)
        rowwise = step_mode == "rowwise"
        opt = optimizer.SGD_Optimizer(
            eng, params, make_rowwise_error_function() if rowwise else make_error_function(),
            trace_depth=3, learning_rate=1e-6, grad_clip=1e-3,
            fused=False if step_mode == "generic" else "auto",
            graph="auto" if step_mode in ("graph", "rowwise") else False,
            speculative=step_mode == "generic")   # (the bench's error function is pure)
        opt.suppress_warnings = True
        for _ in range(args.warmup):
            opt.single_step(None)
        fs = opt._fused_step
        barrier()

        def run(n_steps):
            tests0 = int(fs.tests_total.item()) if fs is not None else 0
            tests_local = 0
            barrier()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                opt.single_step(None)
                if fs is None:
                    tests_local += eng.last_trace["n_tests"]
            barrier()
            dt = time.perf_counter() - t0
            if fs is not None:
                tests_local = int(fs.tests_total.item()) - tests0
            tests_total = float(tests_local)
            per_rank = [dt]
            if world > 1:
                stats = torch.tensor([dt, float(tests_local)], dtype=torch.float64, device="cuda")
                every = [torch.zeros_like(stats) for _ in range(world)]
                torch.distributed.all_gather(every, stats)
                per_rank = [float(e[0].item()) for e in every]
                dt, tests_total = max(per_rank), float(sum(e[1].item() for e in every))
            run.per_rank_s = per_rank
            return dt, tests_total

        def enough_steps():
            # (every rank takes the same count: the slowest rank's estimate)
            t0 = time.perf_counter()
            for _ in range(3):
                opt.single_step(None)
            torch.cuda.synchronize()
            est = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device="cuda")
            if world > 1:
                torch.distributed.all_reduce(est, op=torch.distributed.ReduceOp.MIN)
            return max(args.steps, int(np.ceil(MIN_TIMED_SECONDS / max(float(est.item()), 1e-6))))

        steps = args.steps if exact_steps else enough_steps()
        if profile:
            lib.tfrt_profile_enable(1)
        dt, tests_total = run(steps)
        per_rank_ms = [t / steps * 1e3 for t in run.per_rank_s]
        kernel_ms = []
        if profile:
            import ctypes
            buf = (ctypes.c_float * 8192)()
            kernel_ms = {}
            for kind, name in enumerate(("intersect", "react", "backward", "accumulate")):
                nrec = lib.tfrt_profile_read_kind(kind, buf, 8192)
                kernel_ms[name] = [buf[i] for i in range(max(nrec, 0))]
            lib.tfrt_profile_enable(0)
        steady = None
        if exact_steps:
            n2 = enough_steps()
            dt2, tests2 = run(n2)
            steady = {"steps": n2, "seconds": dt2, "ms_per_step": dt2 / n2 * 1e3,
                      "tests_per_s": tests2 / dt2}
        # work the in-place trace executed in its last step (rank 0's shard)
        executed = None
        st = fs._state if fs is not None else None
        if fs is not None and fs.in_place and st is not None:
            ex = torch.zeros(2, dtype=torch.int64, device="cuda")
            from tensorflowraytrace_amd import ops as _ops
            rc = lib.tfrt_trace3d_executed(st["N"], st["M"], st["P"], st["dt"],
                                           st["ws"].data_ptr(), st["wsb"], ex.data_ptr(),
                                           _ops._stream(ex))
            torch.cuda.synchronize()
            if rc == 0:
                executed = [int(v) for v in ex.tolist()]
        out = dict(dt=dt, steps=steps, tests=tests_total, kernel_ms=kernel_ms, steady=steady,
                   per_rank_ms=per_rank_ms,
                   counts=eng.last_trace["counts"], M=int(system._merged_face_verts.shape[0]),
                   mode=eng._trace_mode(system),
                   visiting=getattr(eng, "_order_cache", None) is not None,
                   in_place=bool(fs.in_place) if fs is not None else False, executed=executed,
                   graph_replays=fs.graph_replays if fs is not None else 0,
                   collective_in_graph=bool(fs.collective_in_graph) if fs is not None else None,
                   capture_error=repr(fs.capture_error) if fs is not None and fs.capture_error
                   else None)
        del eng, system, params, opt
        torch.cuda.empty_cache()
        return out

    per_rank = args.scaling == "weak"
    global_rays = args.rays * world if per_rank else args.rays
    main_leg = timed_leg(global_rays, args.trace_mode, args.step_mode, args.dtype, exact_steps=True)
    side = {}
    if world > 1 and not args.no_extra_legs:
        # the other scaling convention, separately reported (never `value`)
        other_rays = args.rays if per_rank else args.rays * world
        leg = timed_leg(other_rays, args.trace_mode, args.step_mode, args.dtype)
        side["weak_scaling" if not per_rank else "strong_scaling"] = {
            "global_rays": other_rays, "ms_per_step": leg["dt"] / leg["steps"] * 1e3,
            "per_rank_ms": leg["per_rank_ms"], "tests_per_s": leg["tests"] / leg["dt"]}
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
    in_place = bool(main_leg["in_place"]) and visiting
    kernel_name = ("tfrt::k_trace_inplace" if in_place else "tfrt::k_intersect_beam" if visiting else
                   {"all-pairs": "tfrt::k_intersect3d", "group": "tfrt::k_intersect_group"}[mode])
    same_workload = (args.rays == 1_000_000 and world == 1 and args.k_front == 41
                     and args.k_back == 9 and args.dtype == "f32")
    src_hash = _source_hash()

    def pmc_of(tag):
        """Counters of one kernel from profiles/ (separate rocprofv3 --pmc runs of the same step,
        scratch/r05_pmc.sh) -- only for the kernel sources and workload they were collected on."""
        path = os.path.join(ROOT, "profiles", "r05_pmc_%s.json" % tag)
        if not (same_workload and os.path.exists(path)):
            return None, path
        doc = json.load(open(path))
        return (doc if doc.get("source_hash") == src_hash else None), path

    def traffic_of(doc):
        fetch = float(np.mean([p["FETCH_SIZE_KB"] for p in doc["passes"]])) * 1024.0
        write = float(np.mean([p["WRITE_SIZE_KB"] for p in doc["passes"]])) * 1024.0
        return 2.0 * fetch + write      # guide: FETCH_SIZE counts half of wide reads on gfx950

    # ROOFLINE of the dominant kernel, SURVEY.md 8d's quantity: algorithmic bytes per launch
    # (64 B per ray entering a pass + 48 B per face and pass, for the passes ONE launch covers)
    # / that kernel's average launch duration (HIP events on the launch stream, measured live)
    # / 8 TB/s.  With the spatial index the path is bandwidth / latency bound, so the
    # bound is "hbm"; how busy the vector units are is kept beside it as valu_issue_frac.
    roofline = {"kernel": kernel_name, "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBPS,
                "unit": "GB/s", "frac": None, "traffic": None}
    kms = prof_leg["kernel_ms"] if prof_leg is not None else {}
    if kms.get("intersect"):
        ms = np.asarray(kms["intersect"], dtype=np.float64)
        per_launch_passes = P if in_place else 1
        if in_place:
            avg_ms = float(ms.mean())
            per_pass_ms = None
            alg_bytes = float(sum(n_active)) * BYTES_PER_RAY_FWD + P * M * BYTES_PER_FACE
            tests_per_launch = float(sum(n * M for n in n_active))
        else:
            ms = ms[:len(ms) // P * P].reshape(-1, P)
            per_pass_ms = [float(x) for x in ms.mean(axis=0)]
            avg_ms = float(ms.mean())
            alg_bytes = float(np.mean(n_active)) * BYTES_PER_RAY_FWD + M * BYTES_PER_FACE
            tests_per_launch = float(np.mean([n * M for n in n_active]))
        tag = "inplace" if in_place else ("beam" if visiting else mode.replace("-", "_"))
        pmc, pmc_path = pmc_of(tag)
        roofline.update({
            "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9,
            "frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_note": "SURVEY 8d: 64 B x rays entering a pass + 48 B x faces per "
                                      "pass, summed over the %d pass(es) one launch covers"
                                      % per_launch_passes,
            "avg_launch_ms": avg_ms, "launches_per_step": 1 if in_place else P,
            "launches_timed": int(ms.size),
            "timed_by": "HIP events on the launch stream (tfrt_profile_*), eager fused sequence "
                        "run after the timed region with the same steps",
            "tests_decided_per_launch": tests_per_launch,
        })
        if per_pass_ms is not None:
            roofline["per_pass_launch_ms"] = per_pass_ms
        if pmc is not None:
            valu = float(np.mean([p["SQ_INSTS_VALU"] for p in pmc["passes"]]))
            traffic = traffic_of(pmc)
            roofline.update({
                "traffic": traffic,
                "traffic_unit": "HBM-side bytes per launch from the PMC counters: 2 x FETCH_SIZE + "
                                "WRITE_SIZE (the guide's gfx950 correction for wide reads), separate "
                                "--pmc passes",
                "traffic_over_algorithmic": traffic / alg_bytes,
                # how busy the vector units are (NOT the roofline fraction): executed VALU
                # wave-instructions / launch time / (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 op)
                "valu_issue_frac": valu / (avg_ms * 1e-3) / (VALU_SIMDS * CLOCK_HZ / CYCLES_PER_WAVE_OP),
                "valu_wave_instructions_per_launch": valu,
                "salu_per_valu": float(np.mean([p["SQ_INSTS_SALU"] / p["SQ_INSTS_VALU"]
                                                for p in pmc["passes"]])),
                "lds_bank_conflict_ratio": float(np.mean(
                    [p["SQ_LDS_BANK_CONFLICT"] / max(p["SQ_LDS_IDX_ACTIVE"], 1.0)
                     for p in pmc["passes"]])),
                "wait_fraction_of_wave_cycles": float(np.mean(
                    [p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"] for p in pmc["passes"]])),
                "pmc_source": os.path.relpath(pmc_path, ROOT),
            })
        else:
            roofline["pmc_note"] = ("no PMC profile of this kernel source / workload under profiles/ "
                                    "(regenerate with scratch/r05_pmc.sh); counter-derived fields "
                                    "withheld")
        # every hot kernel of the step: live launch time, SURVEY 8d's algorithmic bytes beside the
        # counter traffic
        n_fwd = float(np.mean(n_active))
        # coherent rays: the reverse sweep is ONE launch per step
        n_bwd = len(kms.get("backward") or [])
        n_int = len(kms.get("intersect") or [])
        bwd_one = in_place or n_bwd * P <= n_int + P - 1
        rows = [dict(name=kernel_name, kind="trace" if in_place else "intersect",
                     launches_per_step=1 if in_place else P, bound="hbm", avg_ms=avg_ms,
                     algorithmic_bytes=alg_bytes, achieved_GBps=roofline["achieved"],
                     frac=roofline["frac"], traffic=roofline.get("traffic"),
                     note="all passes in one launch: beam walk, classification, Snell, tape"
                     if in_place else None)]
        for kind, kname, per_step, alg, note in (
                ("react", "tfrt::k_react3d", P, n_fwd * BYTES_PER_RAY_FWD,
                 "SURVEY 8d forward bytes: 64 B per ray entering the pass"),
                ("backward", "tfrt::k_backward_chain" if bwd_one else "tfrt::k_backward3d",
                 1 if bwd_one else P,
                 (float(sum(n_active)) if bwd_one else n_fwd) * BYTES_PER_RAY_BWD,
                 "SURVEY 8d backward bytes: 116 B per ray entering a pass" +
                 (", all passes in the one launch (a lane walks its ray's records back with the child "
                  "gradient in registers; face-gradient terms summed per wavefront in LDS)"
                  if bwd_one else " (one launch per pass)")),
                ("accumulate", "tfrt::k_face_accumulate", 1, float(sum(n_active)) * 40.0,
                 "the stash read once: 36 B of terms + 4 B face index per ray and pass")):
            v = np.asarray(kms.get(kind) or [], dtype=np.float64)
            if not v.size:      # (in-place traces: no reaction launch; coherent rays: no stash)
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
        roofline["kernels_note"] = ("avg_ms: HIP events per launch (eager fused sequence); frac: "
                                    "SURVEY 8d algorithmic bytes / time / 8 TB/s; traffic: 2 x "
                                    "FETCH_SIZE + WRITE_SIZE per launch from profiles/ when collected "
                                    "on these kernel sources, else null")
        roofline["hot_kernels_ms_per_step"] = sum(r["avg_ms"] * r["launches_per_step"] for r in rows)
    dt, tests_total = main_leg["dt"], main_leg["tests"]
    line = {
        "metric": "ray-surface intersection tests/sec (fwd+bwd)",
        "value": tests_total / dt,
        "unit": "tests/s",
        "n_gpus": world,
        "steps": main_leg["steps"],
        "warmup": args.warmup,
        "ms_per_step": dt / main_leg["steps"] * 1e3,
        "per_rank_ms": main_leg["per_rank_ms"],
        "timed_region_s": dt,
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
            + (", rays visited in a coherent (Hilbert) order: wavefronts share one walk"
               + (", all passes in ONE launch with the rays kept in place (k_trace_inplace)"
                  if in_place else " (k_intersect_beam)") if visiting else ""),
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
                      "per-pair test -- ms_per_step is the comparable number, value_executed "
                      "counts the pair tests that ran, and other_legs.all_pairs is the same step "
                      "with every pair through the float32 filter",
        "roofline": roofline,
    }
    if dt < MIN_TIMED_SECONDS:
        line["timed_region_note"] = ("exactly --steps steps were timed: %.1f ms, shorter than %.1f s; "
                                     "`steady` is the same step over a longer loop"
                                     % (dt * 1e3, MIN_TIMED_SECONDS))
    if main_leg.get("steady"):
        line["steady"] = main_leg["steady"]
    if main_leg.get("executed"):
        ex = main_leg["executed"]
        per_s = main_leg["steps"] / dt * world     # (rank 0's shard x ranks)
        line["value_executed"] = {
            "exact_pair_tests_per_s": ex[0] * per_s, "faces_tested_against_bundles_per_s": ex[1] * per_s,
            "exact_pair_tests_per_step": ex[0] * world, "faces_tested_against_bundles_per_step": ex[1] * world,
            "note": "work the forward trace of one step executed: (ray, face) pairs through the exact "
                    "float64 test (tfrt/geometry.py:286-311) and candidate faces tested as triangles "
                    "against a wavefront's bundle (tfrt_trace3d_executed); the reverse sweep re-derives "
                    "one pair per ray and pass"}
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
            leg = timed_leg(args.rays, args.trace_mode, "rowwise", args.dtype)
            legs["rowwise_step"] = {"ms_per_step": leg["dt"] / leg["steps"] * 1e3,
                                    "tests_per_s": leg["tests"] / leg["dt"],
                                    "graph_replays": leg["graph_replays"], "in_place": leg["in_place"],
                                    "note": "the same error function as arbitrary ROW-WISE torch code "
                                            "(optimizer.RowwiseError): evaluated on fixed-shape "
                                            "tensors -- every ray's column + a mask -- inside the "
                                            "step's launch graph, its autograd included; no ray "
                                            "count is read back"}
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
