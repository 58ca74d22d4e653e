"""Bundle compactness of the program order vs the block order (random aperture source)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import tfrt.distributions as d, tfrt.sources as sources
from tensorflowraytrace_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d.seed(5)
a = d.RandomUniformCircle(n, 0.2); d.BasePointTransformation(a, translation=(-10, 0, 0))
b = d.RandomUniformCircle(n, 0.9); d.BasePointTransformation(b)
src = sources.AperatureSource(3, a, b, [575.0], dense=False)
src.update()
rs = src._fields
block = rs.ray_block(torch.float32)
def spread(p):
    out = []
    for rows in ((1, 3), (4, 6)):
        yz = block[rows[0]:rows[1], p.long()][:, :n // 64 * 64].reshape(2, -1, 64)
        ext = yz.max(dim=2)[0] - yz.min(dim=2)[0]
        out.append(float(ext.pow(2).sum(dim=0).sqrt().mean()))
    return out
print("program order      ", spread(rs.order()))
print("block order (axis) ", spread(ops.ray_order(block, None, axis=src.axis_hint())))
print("block order        ", spread(ops.ray_order(block, None)))
