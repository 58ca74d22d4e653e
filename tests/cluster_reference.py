"""The host-side (numpy) k-d face ordering that tfrt_cluster_order replaces: kept as the yardstick
of the device order's quality (tests/test_gpu_order.py).  Test infrastructure only."""
import numpy as np
import torch


def cluster_order_numpy(face_verts, leaf=16, group=128):
    """Permutation of the faces that makes every aligned run of ``leaf`` consecutive entries
    (and of ``group`` entries: the kernels' clusters and superclusters) a compact patch:
    recursive median split of the face centroids along the longest axis of their bounding box,
    the left part a multiple of ``group`` (of ``leaf`` below that), down to parts of <= ``leaf``
    faces.  Pass it as ``cluster_order``.  Unlike a Morton order there are no space-filling-curve jumps inside
    a run, so the clusters' bounding spheres stay small (measured on the cfg4 lens: 10 cluster
    hits per ray instead of 17).  Host-side numpy, once per mesh topology (2 s for 1e6 faces)."""
    fv = face_verts.detach().to("cpu", torch.float64).numpy()
    cent_all = (fv[:, 0:3] + fv[:, 3:6] + fv[:, 6:9]) / 3.0
    # Outsized faces (a target plane behind a fine lens mesh, ...) go last, after the k-d order
    # of the rest: a cluster's bounding sphere contains all its members, so one huge member
    # would make every ray test the 15 small ones that happen to share its cluster.
    size = np.sqrt(((fv.reshape(-1, 3, 3) - cent_all[:, None, :]) ** 2).sum(2)).max(1)
    big = size > 8.0 * np.median(size) if size.size else np.zeros(0, dtype=bool)
    small_idx = np.nonzero(~big)[0]
    cent = cent_all[small_idx]
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
        axis = int(np.argmax(c.max(0) - c.min(0)))
        unit = group if m > group else leaf
        n_left = unit * (((m + unit - 1) // unit) // 2)
        part = np.argpartition(c[:, axis], n_left - 1)
        stack.append((idx[part[n_left:]], at + n_left))
        stack.append((idx[part[:n_left]], at))
    out = np.concatenate([small_idx[out], np.nonzero(big)[0]])
    return torch.as_tensor(out, dtype=torch.int32, device=face_verts.device)


