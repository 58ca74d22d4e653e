import sys, os, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "examples"))
import torch, hexalens
import tfrt.optimizer as optimizer
s = hexalens.build(100000, 0.05)
opt = optimizer.SGD_Optimizer(s["engine"], s["lens"].parameters, s["error_function"], 3, learning_rate=4e-6, grad_clip=1.0)
opt.suppress_warnings = True
for _ in range(3): opt.single_step([s["accumulator"]] * 2 if not isinstance(s["accumulator"], list) else s["accumulator"])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
acc = s["accumulator"]; smo = s["smoother"]
accs = acc if isinstance(acc, (list, tuple)) else [acc] * 2
smos = smo if isinstance(smo, (list, tuple)) else [smo] * 2
for _ in range(10):
    opt.single_step(list(accs))
    for p_, s_ in zip(opt.parameters, smos): opt.smooth(p_, s_)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
