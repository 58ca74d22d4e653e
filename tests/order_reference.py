"""torch restatement of the key function of tfrt_ray_order (csrc/tfrt_order.hip: k_order_xy,
k_order_key) for the tests: same frame, same float32 grid arithmetic, Hilbert index of the cell.
Test infrastructure only."""
import math

import torch


def order_bits(n):
    lg = 0
    while (1 << lg) < n:
        lg += 1
    return min(13, max(4, (lg + 1) // 2))


def hilbert_index(x, y, bits):
    x, y = x.clone(), y.clone()
    d = torch.zeros_like(x)
    n1 = (1 << bits) - 1
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).to(torch.int64)
        ry = ((y & s) > 0).to(torch.int64)
        d += s * s * ((3 * rx) ^ ry)
        swap = ry == 0
        flip = swap & (rx == 1)
        x = torch.where(flip, n1 - x, x)
        y = torch.where(flip, n1 - y, y)
        x, y = torch.where(swap, y, x), torch.where(swap, x, y)
        s >>= 1
    return d


def keys(rays, axis, centre=None):
    """Keys of a (6, N) block for a given axis; centre: the mean end point of the 256 sampled
    rays (what the kernel uses without faces) unless given."""
    r = rays.detach().double().cpu()
    n = r.shape[1]
    s, e = r[:3], r[3:]
    if centre is None:
        idx = (torch.arange(256, dtype=torch.int64) * n) // 256
        es = e[:, idx]
        ok = torch.isfinite(es).all(0)
        centre = es[:, ok].sum(1) / max(int(ok.sum()), 1)
    c = torch.as_tensor(centre, dtype=torch.float64).reshape(3, 1)
    w = torch.as_tensor(axis, dtype=torch.float64)
    w = w / w.norm()
    k = int(torch.argmin(w.abs()))
    ek = torch.zeros(3, dtype=torch.float64)
    ek[k] = 1.0
    a = torch.linalg.cross(w, ek)
    a = a / a.norm()
    b = torch.linalg.cross(w, a)
    d = e - s
    length = d.norm(dim=0)
    good = torch.isfinite(length) & (length > 0)
    u = d / length.clamp(min=1e-300)
    t = ((c - s) * u).sum(0, keepdim=True)
    p = s + t * u - c
    x = (p * a.reshape(3, 1)).sum(0).float()
    y = (p * b.reshape(3, 1)).sum(0).float()
    good &= torch.isfinite(x) & torch.isfinite(y)
    bits = order_bits(n)
    g1 = torch.tensor(float((1 << bits) - 1), dtype=torch.float32)

    def grid(v):
        lo, hi = v[good].min(), v[good].max()
        sc = g1 / (hi - lo) if hi > lo else torch.tensor(0.0)
        return torch.clamp((v - lo) * sc, 0.0, float(g1)).to(torch.int64)

    key = hilbert_index(grid(torch.where(good, x, x[good][0])), grid(torch.where(good, y, y[good][0])), bits)
    return torch.where(good, key, torch.full_like(key, (1 << (2 * bits)) - 1)), bits
