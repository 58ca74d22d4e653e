"""Why does the 60k-ray fused step capture one step later?  prints the left-over counts per step."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch, bench
import tensorflowraytrace_amd as tfa
import tfrt.optimizer as optimizer
tfa.set_device("cuda:0")
eng, system, params = bench.build_scene(60_000, 9, 5, torch.float64)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3,
                              learning_rate=1e-5, grad_clip=1e-3)
opt.suppress_warnings = True
for k in range(8):
    e = float(opt.single_step(None, lr_scale=1.0 - 0.05 * k))
    fs = opt._fused_step
    st = fs._state
    print(k, e, "eager", fs._eager_steps, "replays", fs.graph_replays, "perm", eng._trace_perm is not None,
          "counts", st["counts"].tolist() if st is not None else None,
          "incoh", getattr(eng, "_incoherent_key", None) is not None,
          "visit_all", getattr(eng, "_visit_all_key", None) is not None, flush=True)
