import numpy as np
def kd_order(cent, leaf=16):
    """Permutation of the faces: recursive median split of the centroids along the longest axis
    of their bounding box, left part a multiple of `leaf`, until parts hold <= leaf faces."""
    n = cent.shape[0]
    out = np.empty(n, dtype=np.int64)
    stack = [(np.arange(n), 0)]
    while stack:
        idx, at = stack.pop()
        m = idx.size
        if m <= leaf:
            out[at:at + m] = idx
            continue
        c = cent[idx]
        ax = int(np.argmax(c.max(0) - c.min(0)))
        nl = leaf * (((m + leaf - 1) // leaf) // 2)
        part = np.argpartition(c[:, ax], nl - 1)[:]
        left, right = idx[part[:nl]], idx[part[nl:]]
        stack.append((right, at + nl))
        stack.append((left, at))
    return out
