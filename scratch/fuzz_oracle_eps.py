"""Soups with large epsilons: HIP (all-pairs, hierarchy, coherent) against the CPU oracle.  fuzz_oracle_eps.py FIRST N"""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch
import test_gpu_stress as st
from oracle import tracer
from tensorflowraytrace_amd import ops, _lib
DEV = "cuda:0"
flags = _lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
bad = n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    sc = st._soup(seed)
    for eps in ((1e-6, 0.3, 1e-10), (1e-10, 1.0, 1e-8), (1e-3, 0.02, 1e-4)):
        system = st._oracle_system(sc); system.eps = eps
        src = {k: sc["rays"][i] for i, k in enumerate(st.NAMES)}
        src["ray_id"] = torch.arange(sc["rays"].shape[1], dtype=torch.float64)
        ref = tracer.ray_trace(system, src, max_iterations=3, inherit=("ray_id",), index_type="value",
                               new_ray_length=sc["L"], flags=dict(compile_dead_rays=True, compile_stopped_rays=True))
        fv = sc["P"].to(DEV)
        for mode in ("all-pairs", "hierarchy", "coherent"):
            args = ops.Scene3DArgs(fv, sc["cat"].int().to(DEV), n_in=sc["n_in"].to(DEV), n_out=sc["n_out"].to(DEV),
                                   cluster_order=None if mode == "all-pairs" else ops.cluster_order(fv),
                                   coherent_rays=mode == "coherent")
            args.eps = eps
            out = ops.trace3d(sc["rays"].to(DEV), fv, args, max_passes=3, flags=flags, new_ray_length=sc["L"])
            ok = True
            for cls in ("finished", "active", "stopped", "dead"):
                r = ref[cls]
                n_ref = r["x_start"].shape[0] if r else 0
                ok = ok and out[cls].shape[1] == n_ref
                if ok and n_ref:
                    ok = torch.equal(out[cls + "_id"].cpu().long(), r["ray_id"].long()) and torch.equal(out[cls].cpu(), st._block(r))
            n += 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, eps, mode, flush=True)
print("checked", n, "mismatches", bad)
