import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_reference_optimizer_golden as T
g = np.load(T.GOLD)
for warm in (3, 1, 2):
    opt, lens, eng = T._build(g, "graph")
    acc = [torch.as_tensor(g["accumulator"]), None]; smoother = torch.as_tensor(g["smoother"])
    errs = []
    for step, lr in enumerate(g["lr"]):
        if opt._fused_step is not None: opt._fused_step.graph_warmup = warm
        errs.append(float(opt.single_step(acc, lr_scale=float(lr))))
        if step % 2 == 1: opt.smooth(lens.parameters[0], smoother)
    fs = opt._fused_step
    print("warm", warm, "replays", fs.graph_replays, "cap_err", fs.capture_error, np.round(np.array(errs) - g["errors"], 12))
