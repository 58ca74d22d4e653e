"""Where does the GPU idle inside a step?  Kernel timeline of a few steps (torch profiler),
gaps > 8 us listed with the kernels on either side."""
import sys, os, json, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
from torch.profiler import profile, ProfilerActivity
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
opt.suppress_warnings = True
for _ in range(15): opt.single_step(None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(6): opt.single_step(None)
    torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/trace.json")
ev = json.load(open("/tmp/trace.json"))["traceEvents"]
ks = sorted([e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e], key=lambda e: e["ts"])
print("kernels", len(ks))
t0, t1 = ks[0]["ts"], ks[-1]["ts"] + ks[-1]["dur"]
busy = sum(e["dur"] for e in ks)
print(f"span {(t1-t0)/1e3:.3f} ms for 6 steps = {(t1-t0)/6e3:.3f} ms/step, busy {busy/6e3:.3f} ms/step, idle {(t1-t0-busy)/6e3:.3f} ms/step")
gaps = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(ks, ks[1:]):
    g = b["ts"] - (a["ts"] + a["dur"])
    if g > 8:
        key = (a["name"][:48], b["name"][:48])
        gaps[key][0] += 1; gaps[key][1] += g
for (a, b), (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{tot/6:8.1f} us/step  x{n/6:.1f}  {a}  ->  {b}")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ks:
    agg[e["name"][:110]][0] += 1; agg[e["name"][:110]][1] += e["dur"]
print("--- kernels per step")
for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{tot/6:8.1f} us/step  x{n/6:.1f}  {name}")
