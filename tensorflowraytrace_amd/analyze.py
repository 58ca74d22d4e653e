"""
Utilities to analyse a traced system (tfrt/analyze.py): a batched imaging histogram, an image
inner product, a differentiable-free 2-D histogram on the ray device and the
``DistributionDifferential`` objective built on it.

Not on the per-step hot path; torch ops on whatever device the samples live on.  Plotting in
``imaging_test`` needs matplotlib, which is optional: ``display=True`` without it raises.
"""
import numpy as np
import torch

from . import config


def imaging_test(get_samples, image_range, batch_count=50, bins=128, verbose=True, display=True):
    """Call ``get_samples()`` (-> (n,2) image-plane points) ``batch_count`` times and histogram
    everything (tfrt/analyze.py:9-77).  Returns (h, xedges, yedges, image); ``image`` is the
    matplotlib QuadMesh when ``display`` else None."""
    batches = []
    for i in range(batch_count):
        s = get_samples()
        if isinstance(s, torch.Tensor):
            s = s.detach().cpu().numpy()
        batches.append(np.asarray(s, dtype=np.float64).reshape(-1, 2))
        if verbose:
            print(f"Sampling step {i}/{batch_count}-{100 * i / batch_count:.2f}%.")
    samples = np.concatenate(batches) if batches else np.zeros((0, 2))
    if verbose:
        print(f"final sample shape: {samples.shape}")
        print(f"total rays traced: {samples.shape[0]}")
    if display:
        try:
            import matplotlib.pyplot as plt
        except ImportError as e:
            raise ImportError("imaging_test(display=True) needs matplotlib") from e
        _fig, ax = plt.subplots(1, 1, figsize=(9, 9))
        ax.set_aspect("equal")
        h, xedges, yedges, image = plt.hist2d(samples[:, 0], samples[:, 1], bins=bins,
                                              range=image_range)
        plt.show()
        return h, xedges, yedges, image
    h, xedges, yedges = np.histogram2d(samples[:, 0], samples[:, 1], bins=bins, range=image_range)
    return h, xedges, yedges, None


def inner_product(first, second):
    """Normalised inner product of two images (tfrt/analyze.py:80-88)."""
    first = np.array(first, dtype=np.float64)
    second = np.array(second, dtype=np.float64)
    return float(np.sum(first / np.linalg.norm(first) * (second / np.linalg.norm(second))))


def _fixed_width_bins(values, lo, hi, nbins):
    # tf.histogram_fixed_width_bins: floor((v - lo) / (hi - lo) * nbins), clipped into
    # [0, nbins - 1] -- points outside the range land in the edge bins
    scaled = (values - lo) / (hi - lo) * nbins
    return torch.floor(scaled).clamp_(0, nbins - 1).long()


def histogram2D(x, y, value_range, x_bins=100, y_bins=None, dtype=torch.int32):
    """2-D fixed-width histogram of the points (x, y) on their own device
    (tfrt/analyze.py:94-131).  Returns H of shape (y_bins, x_bins) -- y is the FIRST index, as in
    the reference; out-of-range points are counted in the edge bins.  One bincount instead of
    the reference's per-row map_fn."""
    y_bins = y_bins or x_bins
    x = config.as_f64(x).reshape(-1)
    y = config.as_f64(y).reshape(-1).to(x.device)
    (x_lo, x_hi), (y_lo, y_hi) = [(float(a), float(b)) for a, b in value_range]
    if x.numel() == 0:
        return torch.zeros((y_bins, x_bins), dtype=dtype, device=x.device)
    flat = _fixed_width_bins(y, y_lo, y_hi, y_bins) * x_bins + _fixed_width_bins(x, x_lo, x_hi, x_bins)
    return torch.bincount(flat, minlength=x_bins * y_bins).reshape(y_bins, x_bins).to(dtype)


class DistributionDifferential:
    """Objective that compares the 2-D density of a point set with a goal density
    (tfrt/analyze.py:134-290): sum of squared differences of the L2-normalised histograms, plus
    an optional penalty for points outside the domain (``oob_penalty(distance to the domain
    centre)``, averaged over the penalised points)."""

    def __init__(self, goal, domain, x_bins=50, y_bins=None, oob_penalty=None):
        self._x_bins = x_bins
        self._y_bins = y_bins or x_bins
        try:
            self._domain = domain
            self._x_start, self._x_end = float(domain[0][0]), float(domain[0][1])
            self._y_start, self._y_end = float(domain[1][0]), float(domain[1][1])
        except (IndexError, TypeError) as e:
            raise ValueError("DistributionDifferential: domain must have shape (2, 2).") from e
        if callable(goal):
            if not isinstance(self._x_bins, int) or not isinstance(self._y_bins, int):
                raise TypeError("DistributionDifferential: bin counts must be ints.")
            dev = config.get_device()
            gx = torch.linspace(self._x_start, self._x_end, self._x_bins + 1, dtype=torch.float64,
                                device=dev)
            gy = torch.linspace(self._y_start, self._y_end, self._y_bins + 1, dtype=torch.float64,
                                device=dev)
            gx, gy = (gx[:-1] + gx[1:]) / 2.0, (gy[:-1] + gy[1:]) / 2.0     # bin centres
            self._eval_grid_x, self._eval_grid_y = torch.meshgrid(gx, gy, indexing="xy")
            try:
                goal = goal(self._eval_grid_x, self._eval_grid_y)
            except Exception as e:
                raise ValueError(
                    "DistributionDifferential: goal must be a callable that accepts two arrays "
                    "of points, or a 2D array.") from e
            goal = config.as_f64(goal)
        else:
            goal = config.as_f64(goal)
            if goal.dim() != 2:
                raise ValueError("DistributionDifferential: goal must be 2D.")
            # same assignment as the reference (analyze.py:197): exact for square grids
            self._x_bins, self._y_bins = goal.shape
        self._goal = goal / torch.linalg.norm(goal)
        self._oob_penalty = oob_penalty
        if oob_penalty:
            try:
                oob_penalty(torch.zeros(5, dtype=torch.float64, device=self._goal.device))
            except Exception as e:
                raise ValueError("DistributionDifferential: oob_penalty must be a callable that "
                                 "accepts an array, or None.") from e
        self.saved_histo = None

    def _distance(self, x, y):
        x = x - (self._x_start + self._x_end) / 2.0
        y = y - (self._y_start + self._y_end) / 2.0
        return torch.sqrt(x * x + y * y)

    def __call__(self, x, y):
        x = config.as_f64(x).reshape(-1).to(self._goal.device)
        y = config.as_f64(y).reshape(-1).to(self._goal.device)
        penalty = None
        if self._oob_penalty:
            oob = (x < self._x_start) | (x > self._x_end) | (y < self._y_start) | (y > self._y_end)
            p = self._oob_penalty(self._distance(x[oob], y[oob]))
            # 0/0 -> nan when nothing is out of bounds in the reference; no penalty here
            penalty = (p / p.shape[0]).sum() if p.shape[0] else p.sum()
            x, y = x[~oob], y[~oob]
        histo = histogram2D(x, y, self._domain, x_bins=self._x_bins, y_bins=self._y_bins)
        histo = histo.to(torch.float64)
        norm = torch.linalg.norm(histo)
        histo = histo / norm if float(norm) > 0 else histo
        self.saved_histo = histo
        quality = ((histo - self._goal) ** 2).sum()
        return quality + penalty if penalty is not None else quality
