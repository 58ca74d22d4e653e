import time, torch
g = torch.Generator(device="cpu"); g.manual_seed(1)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
N = 100000
print("threads", torch.get_num_threads())
print("cpu rand f64      %.3f ms" % t(lambda: torch.rand(N, dtype=torch.float64, generator=g)))
print("cpu rand + affine %.3f ms" % t(lambda: 0.5 + 2.0 * torch.rand(N, dtype=torch.float64, generator=g)))
u = torch.rand(N, dtype=torch.float64)
print("h2d pageable      %.3f ms" % t(lambda: u.to("cuda")))
gd = torch.Generator(device="cuda"); gd.manual_seed(1)
print("gpu rand f64      %.3f ms" % t(lambda: torch.rand(N, dtype=torch.float64, generator=gd, device="cuda")))
