"""Soak: many optimiser steps / traces across sizes, dtypes and trace modes; every result finite,
no exception, counts conserved."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "examples"))
import numpy as np, torch, bench
import tfrt.optimizer as optimizer
t0 = time.time()
# 1) bench scene, many steps at several sizes and modes
for N, mode, dt in ((1_000_000, "auto", torch.float32), (333_333, "auto", torch.float32), (50_001, "all-pairs", torch.float32),
                    (200_000, "group", torch.float32), (120_000, "auto", torch.float64), (77_777, "auto", torch.float16)):
    eng, system, params = bench.build_scene(N, 41, 9, dt, accelerate=mode)
    opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-5, grad_clip=1e-3)
    opt.suppress_warnings = True
    steps = 300 if N >= 1_000_000 else 400
    errs = []
    for i in range(steps):
        e = opt.single_step(None)
        if i % 50 == 49: errs.append(float(e))
    torch.cuda.synchronize()
    c = eng.last_trace["counts"]
    assert int(c[0, :4].sum()) == N and all(np.isfinite(errs)), (N, mode, errs)
    assert all(bool(torch.isfinite(p).all()) for p in params)
    print(f"bench scene N={N} {mode} {str(dt)[6:]}: {steps} steps ok, error {errs[0]:.6g} -> {errs[-1]:.6g}, t={time.time()-t0:.0f}s", flush=True)
# 2) hexalens with random sources
import hexalens
errors, s = hexalens.run(ray_count=30000, steps=600, verbose=False)
assert all(np.isfinite(errors)); print(f"hexalens 600 steps ok: {np.mean(errors[:5]):.4g} -> {np.mean(errors[-5:]):.4g}, t={time.time()-t0:.0f}s", flush=True)
# 3) 2-D light guide, 50 bounces, repeated with random rays
import light_guide
for k in range(20):
    eng, system = light_guide.main(sample_count=2000, max_iterations=50, random=True, verbose=False, ray_dtype=torch.float32)
    assert eng.active_rays["x_start"].shape[0] > 0
print(f"light guide 20 runs ok, t={time.time()-t0:.0f}s", flush=True)
