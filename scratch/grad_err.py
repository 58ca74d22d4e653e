"""Parameter gradient of the bench step with float32 ray state against float64 ray state (same
scene, same rays): relative error of the summed gradient, per parameter tensor."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
out = {}
for dt in (torch.float64, torch.float32):
    eng, system, params = bench.build_scene(N, 41, 9, dt)
    opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3, learning_rate=1.0,
                                  grad_clip=1e30, graph=False)
    opt.suppress_warnings = True
    before = [p.detach().clone() for p in params]
    err = float(opt.single_step(None))
    out[dt] = ([(b - p.detach()) / 0.01 for b, p in zip(before, params)], err)
for k, (a, b) in enumerate(zip(out[torch.float64][0], out[torch.float32][0])):
    print(f"parameter {k}: |g32 - g64| max / |g64| max = {float((a - b).abs().max() / a.abs().max()):.3e}, "
          f"relative L2 = {float((a - b).norm() / a.norm()):.3e}")
print("error f64 state %.12e, f32 state %.12e" % (out[torch.float64][1], out[torch.float32][1]))
