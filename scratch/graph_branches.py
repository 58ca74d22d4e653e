"""Do two independent branches of a captured HIP graph run concurrently on this runtime?  Two long
element-wise kernels on forked streams inside one capture; kernel start / end times of the replays from
the torch profiler.  (Experiment for DESIGN 9: filling the end of the trace launch with the reverse
sweep of another half of the rays needs this.)"""
import torch
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
n = 1 << 24
a = torch.rand(n, device=dev); b = torch.rand(n, device=dev)
oa = torch.empty_like(a); ob = torch.empty_like(b)
side = torch.cuda.Stream()


def work(x, out, k):
    # a few hundred microseconds of arithmetic on few workgroups' worth of data is not possible with
    # stock element-wise kernels: they fill the chip.  Long chains instead: each kernel ~20 us.
    y = x
    for _ in range(k):
        y = torch.sin(y)
    out.copy_(y)


for _ in range(3):
    work(a, oa, 4); work(b, ob, 4)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    work(a, oa, 4)
    with torch.cuda.stream(side):
        work(b, ob, 4)
    cur.wait_stream(side)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA],
            key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
overlaps = 0
for i, e in enumerate(ev[:30]):
    print(f"{e.name[:50]:50s} start {e.time_range.start - t0:8.1f} end {e.time_range.end - t0:8.1f} us")
for x, y in zip(ev, ev[1:]):
    if y.time_range.start < x.time_range.end - 1.0:
        overlaps += 1
print("kernels:", len(ev), "overlapping neighbours:", overlaps)
