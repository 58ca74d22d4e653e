"""
Ray-parallel multi-GPU support: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm; "gloo" for CPU tests).

The path shards by rays: every rank holds the full (small) boundary
set and traces a contiguous block of the source rays; the only exchange per optimiser step is
ONE all-reduce(sum) of a flat float64 buffer ``[grad(p_0), ..., grad(p_k), sum(error),
n_error_terms]`` (tens of KB: latency-bound over xGMI, so a single fused buffer and a single
collective).  Because the gradient of a vector error is the gradient of its sum
(optimizer.py:219-220), summing shard gradients is exact; clip / accumulate / SGD / smooth /
constraints then run identically on every rank, so parameters stay bit-identical without a
broadcast.
"""
import datetime
import os
import sys
import threading

import torch
import torch.distributed as dist

# Seconds a collective (and the rendezvous) may take before the process group gives up
# (TFRT_DIST_TIMEOUT).  The exchange of this path is one ~50 KB all-reduce per step: a rank that
# waits two minutes for it is not slow, its peers are gone or a captured collective never ran.
DEFAULT_TIMEOUT_S = 120.0


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rk)))
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        if backend is None:
            backend = os.environ.get("TFRT_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one rank per GPU; the modulo only matters for rehearsals with more ranks than GPUs
            # (TFRT_DIST_BACKEND=gloo), RCCL itself needs distinct devices
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        timeout = float(os.environ.get("TFRT_DIST_TIMEOUT", DEFAULT_TIMEOUT_S))
        if backend == "nccl":
            # (RCCL: a collective that times out aborts the process instead of blocking for ever)
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group(backend=backend, rank=rk, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout))
    return rk, world, local


class Watchdog:
    """Bounded wait around a multi-rank phase: ``with Watchdog(seconds, "what"):`` ends the
    PROCESS with status 124 when the block has not finished in time.  A collective that was captured
    into a HIP graph on one rank and not on another (or whose peer died) does not raise, it hangs
    inside the runtime where no Python exception can reach it -- and a launcher then waits for
    ever.  ``os._exit`` from a timer thread always gets out; nothing is re-executed (a process
    that has touched the GPU must not exec), the launcher sees the status and ends the other
    ranks.  ``seconds <= 0`` or a single process: no timer."""

    def __init__(self, seconds, what, only_distributed=True):
        self.seconds, self.what = float(seconds), what
        self._timer = None
        self._on = self.seconds > 0 and (is_distributed() or not only_distributed)

    def _expire(self):
        try:
            print(f"[tfrt] rank {rank()}: {self.what} did not finish within {self.seconds:.0f} s "
                  f"(a hung collective?) -- exiting with status 124", file=sys.stderr, flush=True)
        finally:
            os._exit(124)

    def __enter__(self):
        if self._on:
            self._timer = threading.Timer(self.seconds, self._expire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def shard_bounds(n, rk=None, world=None):
    """Contiguous block [lo, hi) of n source rays owned by rank rk."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    base, rem = divmod(int(n), int(world))
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


def all_reduce_step(grads, error_sum, error_count):
    """Sum parameter gradients, the error sum and the number of error terms over ranks with a
    single collective.  ``grads``: list of tensors (None allowed -> treated as zeros of the
    matching parameter, supplied as (None, like) tuples).  Returns (grads, error_sum,
    error_count) with the reduced values (error_count: float, or a 0-dim device tensor when
    the buffer lives on a GPU); a no-op for one process."""
    if not is_distributed():
        return grads, error_sum, error_count
    flat = [g.reshape(-1).to(torch.float64) for g in grads]
    dev = flat[0].device if flat else error_sum.device
    tail = torch.stack([error_sum.to(torch.float64).reshape(()).to(dev),
                        torch.as_tensor(float(error_count), dtype=torch.float64, device=dev)])
    buf = torch.cat(flat + [tail])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out, o = [], 0
    for g in grads:
        n = g.numel()
        out.append(buf[o:o + n].reshape(g.shape).to(g.dtype))
        o += n
    # the reduced count stays a device scalar on a GPU: reading it here would block the host on
    # the backward pass + collective once per step
    count = buf[o + 1] if buf.is_cuda else float(buf[o + 1].item())
    return out, buf[o], count
