"""Does ray ORDER matter?  Same bench step with the source rays permuted into Morton order of
their aperture end points (coherent waves) vs the golden-spiral order."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, bench
import tfrt.optimizer as optimizer
from tensorflowraytrace_amd import _lib
import ctypes
N = 1_000_000
def part1by1(v):
    v = v & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
    return v
import tfrt.distributions as distributions
_orig_finish = distributions.CircleBase._finish
MODE = ["spiral"]
def _finish(self):
    # permute the golden-spiral samples (same samples, different order); both circles of the
    # aperture source use the same permutation so ray i still joins start point i and end point i
    if MODE[0] != "spiral":
        n = self._r.shape[0]
        if MODE[0] == "morton":
            yy = self._r * torch.cos(self._theta); zz = self._r * torch.sin(self._theta)
            y = ((yy + 1) * 0.5 * 65535).long().clamp(0, 65535)
            z = ((zz + 1) * 0.5 * 65535).long().clamp(0, 65535)
            perm = torch.argsort(part1by1(y) | (part1by1(z) << 1))
        else:
            g = torch.Generator(device=self._r.device); g.manual_seed(3)
            perm = torch.randperm(n, device=self._r.device, generator=g)
        self._r, self._theta = self._r[perm], self._theta[perm]
    _orig_finish(self)
distributions.CircleBase._finish = _finish
for mode in ("spiral", "morton", "random"):
    MODE[0] = mode
    eng, system, params = bench.build_scene(N, 41, 9, torch.float32)
    opt = optimizer.SGD_Optimizer(eng, params, bench.error_function, trace_depth=3, learning_rate=1e-6, grad_clip=1e-3)
    opt.suppress_warnings = True
    for _ in range(12): opt.single_step(None)
    L = _lib.lib()
    torch.cuda.synchronize(); L.tfrt_profile_enable(1); t = time.perf_counter()
    for _ in range(20): opt.single_step(None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20 * 1e3
    buf = (ctypes.c_float * 4096)(); n = L.tfrt_profile_read(buf, 4096); L.tfrt_profile_enable(0)
    ms = [buf[i] for i in range(n)]
    y = system._amalgamated_sources["y_end"][:3].tolist()
    print(f"{mode:8s} {dt:.3f} ms/step  intersect launches avg {sum(ms)/max(len(ms),1):.4f} ms  first3 {[round(x,3) for x in ms[:3]]} y_end[:3] {y}", flush=True)
