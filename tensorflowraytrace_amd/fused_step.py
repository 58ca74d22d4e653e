"""
The optimiser step as ONE fixed launch sequence (no host read inside, hipGraph-replayable).

``SGD_Optimizer.single_step`` of the reference (tfrt/optimizer.py:187-320) is

    system.update() -> engine.ray_trace(depth) -> error_function(engine) -> tape.gradient
    -> non-finite -> 0, scale, clip, accumulate -> SGD apply

The generic path of this package keeps the user's ``error_function`` arbitrary torch code, which
costs a blocking read of the ray counts (the finished set has a data-dependent length), ~25
stock torch kernels for the error and its reverse, and autograd bookkeeping: ~0.75 ms per step
whatever the ray count -- the part that caps ray-parallel strong scaling.

The error functions of the reference's optimisation scripts all have one form
(dev/hexalens.py:144-168):  ``squared_difference(stack(finished[fields]), goal)`` with ``goal``
a function of fields the finished rays inherit unchanged from their source rays.  ``GoalError``
states that form declaratively; an optimiser given a ``GoalError`` runs ``FusedStep``:

    system.update()                     constraints, parameters -> faces   (torch + tfrt_param_faces_*)
    tfrt_trace3d_forward                into persistent buffers, counts stay on the device
    tfrt_goal_error3d                   error sum (fixed order) + gradient seed 2 (output - goal)
    tfrt_trace3d_backward               -> d error / d faces
    autograd through update()           -> d error / d parameters           (tfrt_param_faces_backward ...)
    [one all-reduce over ray shards]
    tfrt_sgd_process_dev / tfrt_csr_matvec   non-finite -> 0, scale, clip, accumulate, SGD apply

Every launch has step-independent arguments (learning-rate dependent scalars live in a small
device table), so after a few eager steps the sequence is captured once in a HIP graph
(``torch.cuda.CUDAGraph``) and replayed: one graph launch per step.  With ray shards over several
processes the collective splits it in two graphs (RCCL is called eagerly between them).

``GoalError`` is also an ordinary error function (``__call__(engine)``): the generic path gives
the same numbers, which is what the parity tests compare.
"""
import ctypes
import gc

import numpy as np
import torch

from . import _lib, ops
from . import distributed as tdist
from ._lib import RayOut, check

_GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")
_CLASS_FLAGS = (("finished", _lib.COMPILE_FINISHED), ("active", _lib.COMPILE_ACTIVE),
                ("stopped", _lib.COMPILE_STOPPED), ("dead", _lib.COMPILE_DEAD))


class _NotInPlace(RuntimeError):
    """The fixed-shape path of a RowwiseError needs an in-place trace; this step takes the generic
    path instead."""


class GoalError:
    """``error = squared_difference(stack([finished[f] for f in fields], axis=1), goal)``.

    ``goal``: a callable taking the field dict of the SOURCE rays and returning one goal row per
    source ray, shape (N, len(fields)) (or (N,) for a single field); or such a tensor.  Finished
    rays look their row up through the source-ray index the trace carries, which equals
    evaluating the same expression on the fields a finished ray inherits (engine.py:2242-2281).

    Example (dev/hexalens.py:154-157, magnification m)::

        erf = GoalError(("y_end", "z_end"), lambda src: -m * src["object_coords"][:, 1:])
    """

    def __init__(self, fields=("y_end", "z_end"), goal=None, rowwise=False):
        self.fields = tuple(fields)
        # ``rowwise=True`` states that a callable ``goal`` computes row i from ray i's fields alone
        # (dev/hexalens.py:154-157 does: a multiple of the ray's object coordinates).  The fused
        # step may then evaluate it on the source in whatever order it traces the rays, instead of
        # permuting a table made in source order -- for a source re-drawn every step that saves a
        # random gather per step.  Leave it False for goals that depend on a ray's POSITION in the
        # source (a table closed over by the callable, a linspace over the rays, ...).
        self.rowwise = bool(rowwise)
        if not self.fields or any(f not in _GEO3 for f in self.fields):
            raise ValueError(f"GoalError: fields must be taken from {_GEO3}, got {fields!r}")
        if goal is None:
            raise ValueError("GoalError: a goal (callable or tensor) is required")
        if len(set(self.fields)) != len(self.fields):
            raise ValueError(f"GoalError: duplicate fields in {fields!r}")
        self.rows = [_GEO3.index(f) for f in self.fields]
        self.goal = goal
        self._cache = None

    def table(self, src, by_ray=False):
        """(len(fields), N) contiguous float64 goal table of the source set ``src``; with
        ``by_ray`` the (N, len(fields)) contiguous form the callable returns (no transposed copy:
        tfrt_goal_error3d takes either layout)."""
        if hasattr(src, "cache_key"):      # rays made in place (sources.DeviceRaySet): lazy fields
            vals, key = [src], src.cache_key
        else:
            vals = list(src.values()) if hasattr(src, "values") else [src[k] for k in src.keys()]
            key = tuple((id(v), getattr(v, "_version", None)) for v in vals)
        key = (key, bool(by_ray))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        g = self.goal(src) if callable(self.goal) else self.goal
        if hasattr(src, "n_rays"):
            n, dev = src.n_rays, src.device
        else:
            n, dev = src["x_start"].shape[0], src["x_start"].device
        g = torch.as_tensor(g, dtype=torch.float64, device=dev).detach()
        if g.dim() == 1:
            g = g.reshape(-1, 1)
        if tuple(g.shape) != (n, len(self.fields)):
            raise ValueError(f"GoalError: goal has shape {tuple(g.shape)}, expected "
                             f"({n}, {len(self.fields)}) -- one row per source ray")
        table = g.contiguous() if by_ray else g.t().contiguous()
        self._cache = (key, table, vals)     # the keyed tensors stay alive with the key
        return table

    def __call__(self, engine):
        """The same error through the generic path (arbitrary-error-function contract)."""
        fin = engine.finished_rays
        if not bool(fin):
            return torch.zeros((0, len(self.fields)), dtype=torch.float64)
        ids = engine.last_trace["finished_id"].long()
        table = self.table(engine._trace_src)
        out = torch.stack([fin[f] for f in self.fields], dim=1).double()
        return (out - table[:, ids].t()) ** 2


class RowwiseError:
    """An error function of the FINISHED rays that works row by row:
    ``error = fn(rays)`` with ``rays[field]`` the fields of the finished rays (geometry and every
    inherited source field, tfrt/engine.py:2242-2281) and the result one row of error terms per
    ray, shape (n,) or (n, k) -- row i a function of row i of the fields alone.  That is what the
    reference's optimisation scripts compute under their tape (dev/hexalens.py:144-168 is one
    instance; any element-wise torch code qualifies, a mean over the rays does not).

    It is an ordinary ``error_function(engine)``; stating the row-wise contract lets the optimiser
    evaluate ``fn`` on fixed-shape tensors -- every source ray's column, a mask for the rays that
    finished (tfrt_scene3d.in_place == 2) -- so that the step reads no ray count back and its
    launches, ``fn``'s own torch kernels and their autograd included, are captured in one HIP graph
    like a ``GoalError``'s (fused_step.FusedStep).  Columns of rays that did not finish hold finite
    stand-in values (the source ray); their error terms are masked out of the sum and their
    gradients are never read."""

    def __init__(self, fn):
        if not callable(fn):
            raise ValueError("RowwiseError: fn must be callable")
        self.fn = fn
        self.rows = []          # (no built-in goal kernel)

    def __call__(self, engine):
        return self.fn(engine.finished_rays)


class _RowFields:
    """Field mapping handed to a RowwiseError on the fixed-shape path: geometry = the rows of the
    in-place finished block, everything else = the source's own fields in the trace's order."""

    def __init__(self, geo, inherited):
        self._geo, self._inh = geo, inherited

    def __getitem__(self, key):
        if key in self._geo:
            return self._geo[key]
        return self._inh(key)

    def keys(self):
        return list(self._geo.keys())

    def __bool__(self):
        return True


class _HyperTable:
    """(n_parameters, 3) float64 {scale, clip, sgd_learning_rate} on the device, refreshed through
    a ring of pinned host buffers only when a value changes."""

    def __init__(self, n, device, slots=8):
        self.dev = torch.zeros((n, 3), dtype=torch.float64, device=device)
        self._host = [torch.zeros((n, 3), dtype=torch.float64).pin_memory() for _ in range(slots)]
        self._events = [None] * slots
        self._at = 0
        self._current = None

    def set(self, rows):
        if rows == self._current:
            return
        k = self._at
        self._at = (k + 1) % len(self._host)
        if self._events[k] is not None:
            self._events[k].synchronize()     # the copy that last used this buffer is long done
        self._host[k].copy_(torch.tensor(rows, dtype=torch.float64))
        self.dev.copy_(self._host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev.device))
        self._events[k] = ev
        self._current = rows


class FusedStep:
    """Runs ``SGD_Optimizer.single_step`` for a ``GoalError`` as a fixed launch sequence; see the
    module docstring.  One instance per optimizer."""

    # coherent rays: error, gradient seed and reverse sweep as ONE launch (tfrt_trace3d_backward_goal);
    # False: tfrt_goal_error3d + tfrt_trace3d_backward (what a trace in natural order always runs)
    fold_backward = True
    folded_backward = False           # what the last enqueued step did
    in_place = False                  # ... its trace ran all passes in one launch, rays in place
    # Several ranks: capture the RCCL all-reduce inside the step's graph (one graph per step) instead
    # of calling it eagerly between two graphs.  "auto": only with a one-rank group -- the form has
    # been replayed with a one-rank RCCL group (tests/test_gpu_zz_rccl.py), never yet with several
    # RCCL ranks (no multi-GPU box in this round's pool), and a capture that goes wrong ACROSS ranks
    # does not raise, it hangs; the split form is plain eager torch.distributed and costs ~10 us
    # per step.  True: always try (bench.py --collective-in-graph).
    capture_collective = "auto"
    collective_in_graph = False

    def __init__(self, optimizer, graph="auto", graph_warmup=3):
        self.opt = optimizer
        self.graph_mode = graph           # "auto" / True: capture after the warm-up; False: never
        self.graph_warmup = int(graph_warmup)
        self._state = None                # persistent buffers of the current signature
        self._graphs = None               # (signature, graph A, graph B or None)
        self._eager_steps = 0
        self.tests_total = None           # device int64: ray-face tests of all fused steps
        self.steps = 0
        self.graph_replays = 0
        self.capture_error = None
        self.untapped = False             # a parameter reached the faces without boundaries.tap
        self._tap_checks = 0              # eager steps that compared leaf and alias gradients

    # ------------------------------------------------------------------------ eligibility
    @staticmethod
    def eligible(optimizer, args, kwargs):
        eng = optimizer.engine
        if not isinstance(optimizer.error_function, (GoalError, RowwiseError)) or args or kwargs:
            return False
        if isinstance(optimizer.error_function, RowwiseError) and not FusedStep.rowwise_ready(eng):
            return False
        if eng.dimension != 3 or not bool(eng.optical_system):
            return False
        if optimizer.apply_momentum and optimizer.momentum > 0.0:
            return False
        try:
            eng._reaction()
        except RuntimeError:
            return False
        return all(isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float64
                   and p.is_contiguous() for p in optimizer.parameters)

    @staticmethod
    def rowwise_ready(eng):
        """A RowwiseError runs on the fixed-shape path once the engine traces this source in place
        (its rays are ordered and an earlier trace -- of the generic path -- left no wavefront
        over); until then, and whenever that stops holding, the step takes the generic path."""
        return (eng.in_place is not False and eng.coherent is not False and not eng.deterministic
                and getattr(eng, "_visit_all_key", None) is not None
                and not getattr(eng, "_rowwise_off", False))

    # -------------------------------------------------------------------------- buffers
    def _buffers(self, block, fv, P, flags, dt):
        """Persistent outputs / tape / seeds of the trace for this (N, M, P, dtype, flags)."""
        N, M = block.shape[1], fv.shape[0]
        # (the error function's rows are baked into `fields` and into which rows of g_fin are
        # ever written: another GoalError gets fresh, zeroed buffers)
        sig = (N, M, P, dt, flags, str(block.device), tuple(self.opt.error_function.rows))
        st = self._state
        if st is not None and st["sig"] == sig:
            return st
        dev = block.device
        L = _lib.lib()
        wsb = L.tfrt_trace3d_workspace_bytes(N, M, P, dt)
        capN = max(N, 1)
        ints = ops._IntPool(dev, True, _lib.COUNTS_PER_PASS * (P + 1), flags, capN, P)
        caps = {"finished": capN, "active": capN * max(P, 1), "stopped": capN, "dead": capN}
        full, aux, outs = {}, {"counts": ints.counts}, {}
        for name, flag in _CLASS_FLAGS:
            if flags & flag:
                rays = torch.empty((6, caps[name]), dtype=block.dtype, device=dev)
                ids, faces = ints.take(caps[name]), ints.take(caps[name])
                full[name], aux[name + "_id"], aux[name + "_face"] = rays, ids, faces
                outs[name] = ops._ray_out(rays, ids, faces)
            else:
                aux[name + "_id"] = aux[name + "_face"] = None
                outs[name] = ops._ray_out(None, None, None)
        # the rays still active after the last pass are not copied out by a fused step
        aux["unfinished"] = torch.empty((6, 0), dtype=block.dtype, device=dev)
        aux["unfinished_id"] = torch.empty(0, dtype=torch.int32, device=dev)
        gws = L.tfrt_goal_error3d_workspace_bytes(capN)
        st = dict(
            sig=sig, N=N, M=M, P=P, dt=dt, flags=flags, capN=capN, full=full, aux=aux, outs=outs,
            ws=torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev), wsb=wsb,
            counts=ints.counts, ints=ints,
            g_fin=torch.zeros((6, capN), dtype=torch.float64, device=dev),
            g_fv=torch.zeros((max(M, 1), 9), dtype=torch.float64, device=dev),
            err=torch.zeros(3, dtype=torch.float64, device=dev),
            goal_ws=torch.zeros(max(gws, 1), dtype=torch.uint8, device=dev), gws=gws,
            fields=(ctypes.c_int32 * 6)(*(self.opt.error_function.rows + [0] * 6)[:6]),
            no_outs={name: ops._ray_out(None, None, None) for name, _ in _CLASS_FLAGS},
        )
        self._state = st
        self._graphs = None
        if self.tests_total is None or self.tests_total.device != dev:
            self.tests_total = torch.zeros(1, dtype=torch.int64, device=dev)
        return st

    # ---------------------------------------------------------------- the launch sequence
    def _enqueue_gradient(self):
        """update -> trace -> error -> reverse sweep -> parameter gradients.  Returns
        (grads, error tensor {sum, terms, mean}).  Nothing here waits for the device."""
        opt, eng = self.opt, self.opt.engine
        system = eng.optical_system
        eng.clear_ray_history()
        from . import boundaries
        with boundaries.collect_taps() as tap_log:
            system.update()
        src = eng._source_set()
        if not src:
            raise RuntimeError("FusedStep: the optical system has no source rays")
        erf = opt.error_function
        block, scene, fv = eng._trace_inputs(src)
        perm = eng._trace_perm
        if isinstance(erf, RowwiseError):
            return self._enqueue_rowwise(erf, src, block, scene, fv, perm, tap_log)
        goal_by_ray = False       # (N, fields) rows instead of (fields, N) columns
        if perm is None:
            goal = erf.table(src)
        else:
            # coherent order: the trace runs over src[perm] (its ray ids are positions in that
            # order), so the goal rows go along; the ray sets are restored when somebody asks
            # (keyed by the goal TABLE, not by the geometry the order was made from: a goal may
            # read any source field, and erf.table() is itself cached by the versions of all of
            # them -- a field changed in place re-evaluates it and, through the key, re-gathers)
            rowwise = bool(erf.rowwise and callable(erf.goal) and hasattr(src, "permuted"))
            base = None if rowwise else erf.table(src)
            gkey = (id(erf), src.cache_key if rowwise else id(base), id(perm), rowwise)
            cached = getattr(self, "_goal_perm", None)
            if cached is None or cached[0] != gkey:
                if rowwise:
                    # made in the trace's order, left in the layout the callable returns
                    rows, by_ray = erf.table(src.permuted(perm), by_ray=True), True
                else:
                    rows, by_ray = ops.gather_rows(base, perm), False
                cached = self._goal_perm = (gkey, rows, perm, by_ray, base)   # (holds `base`: its id stays its own)
            goal, goal_by_ray = cached[1], cached[3]
        P, flags = int(opt.trace_depth), eng._flags() | _lib.COMPILE_FINISHED
        dt = ops._DT[block.dtype]
        fvc = fv.detach()
        if fvc.dtype != torch.float64 or not fvc.is_contiguous():
            raise RuntimeError("FusedStep: merged faces must be contiguous float64")
        st = self._buffers(block, fvc, P, flags, dt)
        sink = getattr(self, "_err_sink", None)
        if tdist.is_distributed() and sink is not None and st["err"].data_ptr() != sink.data_ptr():
            st["err"] = sink             # (the error sums land in the collective's buffer)
        L = _lib.lib()
        stream = ops._stream(block)
        sc = scene.struct(fvc)
        o = st["outs"]
        need_back = bool(fv.requires_grad and st["M"] > 0)
        # coherent rays: error, gradient seed and the whole reverse sweep are ONE launch
        # (tfrt_trace3d_backward_goal); the face-gradient block it accumulates into is cleared by
        # the trace's set-up launch
        folded = bool(self.fold_backward and need_back and sc.coherent_rays
                      and not sc.deterministic and 1 <= P <= 8)
        self.folded_backward = folded
        # all passes in one launch, rays in place (tfrt_scene3d.in_place): the folded reverse sweep
        # needs no ray set, so none is compacted here -- _publish_lazily does it if somebody asks
        inplace = bool(folded and sc.in_place and st["N"] >= 64 and P >= 1)
        self.in_place = inplace
        if not inplace:
            sc.in_place = 0
        if folded:
            sc.clear_buffer, sc.clear_count = st["g_fv"].data_ptr(), st["g_fv"].numel()
        fo = st["no_outs"] if inplace else o
        try:
            check(L.tfrt_trace3d_forward(
                ops._p(block), block.shape[1], st["N"], ctypes.byref(sc), float(eng.new_ray_length),
                float(eng.dead_ray_length or 0.0), P, dt, flags, ctypes.byref(fo["finished"]),
                ctypes.byref(fo["active"]), ctypes.byref(fo["stopped"]), ctypes.byref(fo["dead"]),
                None, None,      # (the rays still active after the last pass are not copied out)
                ops._p(st["counts"]), ops._p(st["ws"]), st["wsb"], stream), "tfrt_trace3d_forward")
        finally:
            # (the struct is cached by the scene: nobody else's trace clears our block)
            sc.clear_buffer, sc.clear_count = None, 0
        fin = st["full"]["finished"]
        # error + gradient seed; the same launch clears the face-gradient block the reverse sweep
        # accumulates into and adds the trace's test count to the running total
        # (one rank: nothing on the device waits for the error sum, so its second stage is left
        # to the parameter update's launch -- _enqueue_apply --; with several ranks the sum goes
        # into the collective and is finished here)
        goal_args = (
            ops._p(fin), st["capN"], ops._p(st["aux"]["finished_id"]), dt, ops._p(st["counts"]), P,
            st["fields"], len(erf.rows), ops._p(goal),
            1 if goal_by_ray else goal.shape[1], goal.shape[1] if goal_by_ray else 1,
            ops._p(st["g_fin"]),
            ops._p(st["err"]), ops._p(st["g_fv"]) if need_back else None,
            st["g_fv"].numel() if need_back else 0, ops._p(self.tests_total),
            ops._p(st["goal_ws"]), st["gws"])
        self._goal_pending = None
        if folded:
            if "chain_ws" not in st:
                cwb = L.tfrt_trace3d_backward_goal_workspace_bytes(st["N"])
                st["chain_ws"] = (torch.zeros(max(cwb, 1), dtype=torch.uint8, device=block.device), cwb)
            pending = _lib.GoalPending()
            check(L.tfrt_trace3d_backward_goal(
                ops._p(block), block.shape[1], st["N"], ctypes.byref(sc),
                float(eng.new_ray_length), float(eng.dead_ray_length or 0.0), P, dt,
                ctypes.byref(o["finished"]), st["fields"], len(erf.rows), ops._p(goal),
                1 if goal_by_ray else goal.shape[1], goal.shape[1] if goal_by_ray else 1,
                ops._p(st["err"]), ops._p(self.tests_total), ops._p(st["chain_ws"][0]),
                st["chain_ws"][1], ctypes.byref(pending), None, 0, None, 0, None, 0,
                ops._p(st["g_fv"]), None, ops._p(st["counts"]), ops._p(st["ws"]), st["wsb"],
                stream), "tfrt_trace3d_backward_goal")
            if tdist.is_distributed():      # (the sum goes into the collective: finished here)
                check(L.tfrt_goal_finish(ctypes.byref(pending), stream), "tfrt_goal_finish")
            else:
                self._goal_pending = (pending, stream)
        elif tdist.is_distributed():
            check(L.tfrt_goal_error3d(*goal_args, stream), "tfrt_goal_error3d")
        else:
            pending = _lib.GoalPending()
            check(L.tfrt_goal_error3d_deferred(*goal_args, ctypes.byref(pending), stream),
                  "tfrt_goal_error3d_deferred")
            self._goal_pending = (pending, stream)
        grads = [None] * len(opt.parameters)
        if need_back:
            if not folded:
                check(L.tfrt_trace3d_backward(
                    ops._p(block), block.shape[1], st["N"], ctypes.byref(sc),
                    float(eng.new_ray_length), float(eng.dead_ray_length or 0.0), P, dt,
                    ops._p(st["g_fin"]), st["capN"], None, 0, None, 0, None, 0, ops._p(st["g_fv"]),
                    None, ops._p(st["counts"]), ops._p(st["ws"]), st["wsb"], stream),
                    "tfrt_trace3d_backward")
            grads = self._parameter_gradients(fv, st, tap_log)
        self._publish_lazily(st, src, P, flags, perm, (block, float(eng.dead_ray_length or 0.0))
                             if inplace else None)
        # (publish() inverts `perm` when it runs: a replayed graph re-orders a re-drawn source into
        # the same tensor behind Python's back)
        return grads, st["err"]

    def _parameter_gradients(self, fv, st, tap_log):
        """d error / d parameters from d error / d faces (st["g_fv"]) through update()'s graph."""
        opt = self.opt
        grads = [None] * len(opt.parameters)
        if True:
            # Inside a graph capture: differentiate w.r.t. the aliases update() read the parameters
            # through (see boundaries.tap), never w.r.t. the leaves.  Outside a capture the leaf is
            # differentiated too -- its gradient is the total whatever route update() took -- and
            # the first eager steps compare the two: a parameter that reaches the faces (partly)
            # without an alias (a custom _update reading the leaf) must never be captured.
            capturing = torch.cuda.is_current_stream_capturing()
            inputs, owner = [], []
            for i, p in enumerate(opt.parameters):
                taps = tap_log.get(id(p), [])
                if not taps:
                    self.untapped = True
                checked = self._tap_checks >= self.graph_warmup and not self.untapped
                if (capturing or checked) and taps:
                    inputs.extend(taps)
                    owner.extend([i] * len(taps))
                else:
                    inputs.extend(taps + [p])
                    owner.extend([i] * len(taps) + [-1 - i])
            with torch.autograd.set_multithreading_enabled(False):
                got = torch.autograd.grad([fv], inputs, grad_outputs=[st["g_fv"]],
                                          allow_unused=True)
            total = {}
            for i, g in zip(owner, got):
                if g is None:
                    continue
                if i < 0:
                    total[-1 - i] = g
                else:
                    grads[i] = g if grads[i] is None else grads[i] + g
            for i, g in total.items():
                if not self.untapped and self._tap_checks < self.graph_warmup:
                    # (one host read per parameter, first steps only.  Relative to the gradient's
                    # largest entry: the aliases are summed in another order than autograd's own
                    # accumulation, entries near zero may differ by more than any relative bound)
                    same = grads[i] is not None and bool(
                        torch.isfinite(g).eq(torch.isfinite(grads[i])).all()) and float(
                        (torch.nan_to_num(grads[i] - g, nan=0.0, posinf=0.0, neginf=0.0)).abs().max()
                    ) <= 1e-9 * float(torch.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0).abs().max())
                    if not same:
                        self.untapped = True
                        self.capture_error = RuntimeError(
                            f"FusedStep: parameter {i} reaches the faces without boundaries.tap "
                            "(its leaf gradient differs from the sum over its aliases): the step "
                            "is never captured in a launch graph")
                grads[i] = g
            if not capturing:
                self._tap_checks += 1
        return grads

    def _enqueue_rowwise(self, erf, src, block, scene, fv, perm, tap_log):
        """update (done) -> in-place trace with the finished rows at the rays' own columns ->
        ``erf.fn`` on fixed-shape tensors + its autograd (torch) -> reverse sweep -> parameter
        gradients.  No ray count is read; everything is capturable."""
        opt, eng = self.opt, self.opt.engine
        P, flags = int(opt.trace_depth), eng._flags() | _lib.COMPILE_FINISHED
        dt = ops._DT[block.dtype]
        fvc = fv.detach()
        if fvc.dtype != torch.float64 or not fvc.is_contiguous():
            raise RuntimeError("FusedStep: merged faces must be contiguous float64")
        st = self._buffers(block, fvc, P, flags, dt)
        N, M, dev = st["N"], st["M"], block.device
        L = _lib.lib()
        stream = ops._stream(block)
        sc = scene.struct(fvc)
        if not (scene.in_place and sc.coherent_rays
                and L.tfrt_trace3d_in_place(ctypes.byref(sc), N, P) == 1):
            eng._rowwise_off = True      # (this source is not traced in place: generic path)
            raise _NotInPlace()
        if "rows" not in st:
            st["rows"] = torch.zeros((6, st["capN"]), dtype=block.dtype, device=dev)
            st["row_face"] = torch.full((st["capN"],), -1, dtype=torch.int32, device=dev)
            st["row_passes"] = torch.zeros(st["capN"], dtype=torch.int32, device=dev)
            st["row_out"] = ops._ray_out(st["rows"], st["row_passes"], st["row_face"])
        need_back = bool(fv.requires_grad and M > 0)
        self.folded_backward, self.in_place = False, True
        sc.in_place = 2
        if need_back:
            sc.clear_buffer, sc.clear_count = st["g_fv"].data_ptr(), st["g_fv"].numel()
        no = st["no_outs"]
        try:
            check(L.tfrt_trace3d_forward(
                ops._p(block), block.shape[1], N, ctypes.byref(sc), float(eng.new_ray_length),
                float(eng.dead_ray_length or 0.0), P, dt, flags, ctypes.byref(st["row_out"]),
                ctypes.byref(no["active"]), ctypes.byref(no["stopped"]), ctypes.byref(no["dead"]),
                None, None, ops._p(st["counts"]), ops._p(st["ws"]), st["wsb"], stream),
                "tfrt_trace3d_forward")
            # the user's error function, row by row on every source ray's column
            leaf = st["rows"].detach().requires_grad_(True)
            geo = {name: leaf[k] for k, name in enumerate(_GEO3)}
            if perm is None:
                inherited = lambda key: src[key]                       # noqa: E731
            else:
                cache = st.setdefault("inherited", {})

                def inherited(key, cache=cache):
                    v = src[key]
                    hit = cache.get(key)
                    if hit is None or hit[0] is not v or hit[1] is not perm or hit[2] != v._version:
                        hit = cache[key] = (v, perm, v._version, v.index_select(0, perm.long()))
                    return hit[3]
            e = erf.fn(_RowFields(geo, inherited))
            if e.dim() == 1:
                e = e.reshape(-1, 1)
            if e.shape[0] != N:
                raise RuntimeError(f"RowwiseError: fn returned {tuple(e.shape)}, expected one row "
                                   f"per ray ({N})")
            mask = st["row_face"][:N] >= 0
            err_sum = torch.where(mask.unsqueeze(1), e.double(), torch.zeros((), dtype=torch.float64,
                                                                           device=dev)).sum()
            terms = mask.sum().double() * e.shape[1]
            with torch.no_grad():
                st["err"][0] = err_sum.detach()
                st["err"][1] = terms
                st["err"][2] = torch.where(terms > 0, err_sum.detach() / torch.clamp(terms, min=1.0),
                                           torch.full_like(terms, float("nan")))
                self.tests_total += st["row_passes"][:N].sum() * M
            grads = [None] * len(opt.parameters)
            if need_back and err_sum.requires_grad:
                with torch.autograd.set_multithreading_enabled(False):
                    g_rows, = torch.autograd.grad(err_sum, [leaf])
                g64 = g_rows.to(torch.float64).contiguous()
                check(L.tfrt_trace3d_backward(
                    ops._p(block), block.shape[1], N, ctypes.byref(sc), float(eng.new_ray_length),
                    float(eng.dead_ray_length or 0.0), P, dt, ops._p(g64), g64.shape[1], None, 0,
                    None, 0, None, 0, ops._p(st["g_fv"]), None, ops._p(st["counts"]),
                    ops._p(st["ws"]), st["wsb"], stream), "tfrt_trace3d_backward")
                grads = self._parameter_gradients(fv, st, tap_log)
        finally:
            sc.in_place = 1 if scene.in_place else 0      # (the struct is cached by the scene)
            sc.clear_buffer, sc.clear_count = None, 0
        self._goal_pending = None
        self._publish_lazily(st, src, P, flags, perm, (block, float(eng.dead_ray_length or 0.0)))
        return grads, st["err"]

    def _publish_lazily(self, st, src, P, flags, perm=None, inplace=None):
        eng = self.opt.engine
        eng._trace_src = src
        eng._trace_sig = (src.n_rays if hasattr(src, "n_rays") else src["x_start"].shape[0], P, flags)
        full, aux = st["full"], st["aux"]
        self._last_perm, self._last_inplace = perm, inplace

        def publish():
            if inplace is not None:
                # the trace ran in place and compacted nothing: the ray sets are gathered from its
                # tape now, into the reference's (per-pass, stable) order of the traced rays
                # (with a coherent order: in the SOURCE's order, through the inverse of the order)
                block, dead_len = inplace
                o = st["outs"]
                check(_lib.lib().tfrt_trace3d_compact(
                    ops._p(block), block.shape[1], st["N"], dead_len, P, st["dt"], flags,
                    ctypes.byref(o["finished"]), ctypes.byref(o["active"]),
                    ctypes.byref(o["stopped"]), ctypes.byref(o["dead"]), None, None,
                    ops._p(st["counts"]), st["M"],
                    None if perm is None else ops._p(ops.inverse_order(perm)), ops._p(st["ws"]),
                    st["wsb"], ops._stream(block)), "tfrt_trace3d_compact")
                return ops._finish_trace(dict(full), dict(aux), P, None)
            out = ops._finish_trace(dict(full), dict(aux), P, None)
            return out if perm is None else ops.restore_order(out, perm)
        eng._pending_trace = publish

    def _enqueue_apply(self, grads, accumulators):
        """non-finite -> 0, scale, clip, accumulate, SGD apply (optimizer.py:223-257, 316) with
        the step-dependent scalars read from the device table."""
        opt = self.opt
        L = _lib.lib()
        k = len(grads)
        if 1 < k <= 8 and all(a is None for a in accumulators) and \
                len({ops._stream(p).value for p in opt.parameters}) == 1:
            # plain SGD on every parameter tensor: one launch for all of them (the device table
            # holds the three scalars of parameter i at offset 3 i)
            gp = (ctypes.c_void_p * k)(*[g.data_ptr() for g in grads])
            pp = (ctypes.c_void_p * k)(*[p.data_ptr() for p in opt.parameters])
            nn = (ctypes.c_int64 * k)(*[g.numel() for g in grads])
            pending, self._goal_pending = getattr(self, "_goal_pending", None), None
            with torch.no_grad():
                if pending is not None and pending[1].value == ops._stream(opt.parameters[0]).value:
                    check(L.tfrt_sgd_process_multi_finish(
                        k, gp, None, pp, nn, ctypes.c_void_p(self._hyper.dev.data_ptr()),
                        ctypes.byref(pending[0]), ops._stream(opt.parameters[0])),
                        "tfrt_sgd_process_multi_finish")
                else:
                    if pending is not None:
                        check(L.tfrt_goal_finish(ctypes.byref(pending[0]), pending[1]),
                              "tfrt_goal_finish")
                    check(L.tfrt_sgd_process_multi(k, gp, None, pp, nn,
                                                   ctypes.c_void_p(self._hyper.dev.data_ptr()),
                                                   ops._stream(opt.parameters[0])),
                          "tfrt_sgd_process_multi")
            return
        pending, self._goal_pending = getattr(self, "_goal_pending", None), None
        if pending is not None:      # (other update paths: the launch tfrt_goal_error3d would have made)
            check(L.tfrt_goal_finish(ctypes.byref(pending[0]), pending[1]), "tfrt_goal_finish")
        for i, (g, p) in enumerate(zip(grads, opt.parameters)):
            hyper = ctypes.c_void_p(self._hyper.dev.data_ptr() + 24 * i)
            stream = ops._stream(p)
            with torch.no_grad():
                if accumulators[i] is None:
                    check(L.tfrt_sgd_process_dev(ops._p(g), None, ops._p(p), g.numel(), _lib.F64,
                                                 hyper, stream), "tfrt_sgd_process_dev")
                else:
                    processed = torch.empty_like(g)
                    check(L.tfrt_sgd_process_dev(ops._p(g), ops._p(processed), None, g.numel(),
                                                 _lib.F64, hyper, stream), "tfrt_sgd_process_dev")
                    acc = opt._matrix_product(opt._acc_cache, i, accumulators[i], processed)
                    apply_row = ctypes.c_void_p(self._hyper_apply.dev.data_ptr() + 24 * i)
                    check(L.tfrt_sgd_process_dev(ops._p(acc.contiguous()), None, ops._p(p),
                                                 acc.numel(), _lib.F64, apply_row, stream),
                          "tfrt_sgd_process_dev")

    def _fix_grads(self, grads):
        opt = self.opt
        out = []
        for g, p in zip(grads, opt.parameters):
            if g is None:
                if not opt.suppress_warnings:
                    print("Warning: SGD_Optimizer.process_gradient encountered a possible issue:  "
                          "The gradient was likely None, which can mean that the error does not "
                          "depend on it.  The gradient will be set to zero and future instances "
                          "of this message will be suppressed.")
                    opt.suppress_warnings = True
                g = torch.zeros_like(p)
            out.append(g.contiguous())
        return out

    def _sequence(self, accumulators, world):
        """The whole step; with several ranks the collective sits between the two halves."""
        if world > 1:
            # one persistent buffer [grad p_0 ... grad p_k, sum of errors, terms, -]: the face
            # updates' reverse kernels write the parameter gradients straight into it (ops.GradSink)
            # and the error kernel its sums, so the collective needs no concatenation before and
            # no slicing after
            opt = self.opt
            n = sum(p.numel() for p in opt.parameters)
            flat = getattr(self, "_flat_buf", None)
            if flat is None or flat.numel() != n + 3 or flat.device != opt.parameters[0].device:
                flat = self._flat_buf = torch.zeros(n + 3, dtype=torch.float64,
                                                    device=opt.parameters[0].device)
            self._err_sink = flat[n:]
            with ops.GradSink(flat[:n]) as sink:
                grads, err = self._enqueue_gradient()
            fixed = self._fix_grads(grads)
            if (err.data_ptr() == flat[n:].data_ptr() and sink.at == n
                    and all(sink.holds(g) for g in fixed)):
                self._flat, self._flat_views = flat, True
            else:       # (a gradient from somewhere else: a parameter with several aliases, ...)
                self._flat = torch.cat([g.reshape(-1) for g in fixed] + [err[:2]])
                self._flat_views = False
            return fixed
        grads, err = self._enqueue_gradient()
        grads = self._fix_grads(grads)
        self._enqueue_apply(grads, accumulators)
        self._err_view = err
        return grads

    def _after_reduce(self, grads, accumulators):
        flat = self._flat
        if self._flat_views:             # the gradients ARE slices of the reduced buffer
            red, o = grads, flat.numel() - 3
        else:
            o, red = 0, []
            for g in grads:
                red.append(flat[o:o + g.numel()].reshape(g.shape))
                o += g.numel()
        self._enqueue_apply(red, accumulators)
        mean = torch.where(flat[o + 1] > 0, flat[o] / torch.clamp(flat[o + 1], min=1.0),
                           torch.full_like(flat[o], float("nan")))
        self._err_view = torch.stack([flat[o], flat[o + 1], mean])

    # ------------------------------------------------------------------------------ step
    def _hyper_rows(self, lr_scale):
        opt = self.opt
        rows, apply_rows = [], []
        for i in range(len(opt.parameters)):
            scale = float(lr_scale * opt.individual_lr[i] * opt.learning_rate)
            clip = float(opt.grad_clip if opt.clip_mode == "common" else
                         opt.individual_lr[i] * opt.clip_scale * opt.learning_rate * lr_scale)
            rows.append((scale, clip, float(opt.sgd_learning_rate)))
            apply_rows.append((1.0, float("inf"), float(opt.sgd_learning_rate)))
        return tuple(rows), tuple(apply_rows)

    def _signature(self, accumulators):
        """Everything a captured graph has baked in and the caller could have changed."""
        opt, eng = self.opt, self.opt.engine
        src = eng.optical_system._amalgamated_sources
        if src and hasattr(src, "identity"):
            src_id = src.identity        # (rays re-drawn in place: the buffers stay)
        else:
            src_id = tuple(id(src[f]) for f in _GEO3) if src else ()
        return (tuple(id(a) for a in accumulators), tuple(p.data_ptr() for p in opt.parameters),
                src_id, int(opt.trace_depth),
                eng._flags(), eng.new_ray_length, eng.dead_ray_length, eng._trace_mode(),
                id(opt.error_function), id(getattr(opt.error_function, "goal", None)),
                getattr(opt.error_function, "fields", None),
                tdist.world_size(), eng.optical_system.scene_signature(), bool(eng.deterministic),
                id((getattr(eng, "_order_cache", None) or (None, None, None))[2]),
                getattr(eng, "_visit_all_key", None) is not None, eng.in_place)

    def step(self, accumulators, lr_scale):
        """One optimiser step.  Returns the error tensor {sum, n_terms, mean} (device)."""
        opt = self.opt
        dev = opt.parameters[0].device
        world = 2 if tdist.is_distributed() else 1      # > 1: the collective splits the sequence
        if getattr(self, "_hyper", None) is None:
            self._hyper = _HyperTable(len(opt.parameters), dev)
            self._hyper_apply = _HyperTable(len(opt.parameters), dev)
        rows, apply_rows = self._hyper_rows(lr_scale)
        self._hyper.set(rows)
        self._hyper_apply.set(apply_rows)
        self.steps += 1

        sig = self._signature(accumulators)
        want_graph = self.graph_mode in ("auto", True) and self.capture_error is None
        if want_graph and self._graphs is not None and self._graphs[0] == sig:
            return self._replay()
        if (want_graph and self._eager_steps >= self.graph_warmup and self._stable(sig)
                and not self.untapped):
            return self._capture(sig, accumulators, world)   # (runs this step, then tries to capture)
        self._eager(accumulators, world)
        self._eager_steps += 1
        self._last_sig = sig
        if (self._eager_steps == 3 and self._state is not None
                and getattr(opt.engine, "_trace_perm", None) is not None):
            # (ONE host read, early on: did the sorted source leave wavefronts to the grouped
            # kernel?  Many: coherent="auto" goes back to natural order for this source)
            opt.engine._note_left_over(int(self._state["counts"][-1]), self._state["P"])
            self._last_sig = self._signature(accumulators)   # (what the note changes is no instability)
        return self._err_view

    def _eager(self, accumulators, world):
        grads = self._sequence(accumulators, world)
        if world > 1:
            torch.distributed.all_reduce(self._flat, op=torch.distributed.ReduceOp.SUM)
            self._after_reduce(grads, accumulators)

    def _replay(self):
        _, ga, gb, _grads = self._graphs
        ga.replay()
        if gb is not None:
            torch.distributed.all_reduce(self._flat, op=torch.distributed.ReduceOp.SUM)
            gb.replay()
        self.graph_replays += 1
        # (a replay re-draws a device-made source behind Python's back: what the source and its
        # distributions had materialised for an earlier draw is stale now)
        for source in getattr(self.opt.engine.optical_system, "_sources", ()):
            note = getattr(source, "note_external_update", None)
            if note is not None:
                note()
        self._republish()
        return self._err_view

    def _stable(self, sig):
        return getattr(self, "_last_sig", None) == sig

    def _republish(self):
        st = self._state
        eng = self.opt.engine
        eng.clear_ray_history()
        self._publish_lazily(st, eng._trace_src, st["P"], st["flags"],
                             getattr(self, "_last_perm", None), getattr(self, "_last_inplace", None))

    def _capture(self, sig, accumulators, world):
        """Run THIS step eagerly on the capture stream, then capture the sequence there (capturing
        executes nothing).  The eager run on that stream matters: autograd remembers the stream
        on which a leaf's gradient accumulator was created; accumulators left over from steps on
        the caller's stream would make the backward inside the capture fork to that stream, and
        ending the capture then crashes inside the HIP runtime."""
        dev = self.opt.parameters[0].device
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream(dev)
        side, cur = self._stream, torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._eager(accumulators, world)
            err_now = self._err_view.clone()
        cur.wait_stream(side)
        self._last_sig = sig
        self._eager_steps += 1
        # the step itself is done; capturing is an optimisation of the following ones: anything in
        # update() that a capture does not allow (a host->device copy, a blocking read) turns it
        # off for this optimizer and the steps go on eagerly
        # No cyclic garbage collection while a stream captures: the collector may run at any
        # allocation, and an unreachable cycle that holds a CUDAGraph or an event of an earlier
        # optimiser (e.g. the previous FusedStep <-> SGD_Optimizer pair) is then destroyed inside
        # the capture -- the HIP runtime refuses the destroy call and the process aborts.
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            torch.cuda.synchronize(dev)
            # (the device is idle: one read tells whether the visiting-order trace of the step just
            # run left wavefronts to the grouped kernel; if not, the captured sequence omits it)
            if self._state is not None:
                self.opt.engine._note_left_over(int(self._state["counts"][-1]), self._state["P"])
            sig = self._signature(accumulators)
            pool = torch.cuda.graph_pool_handle()
            ga, gb, grads = None, None, None
            want = self.capture_collective
            if want == "auto":
                want = torch.distributed.is_initialized() and torch.distributed.get_world_size() == 1
            if world > 1 and want and torch.distributed.get_backend() == "nccl":
                # RCCL collectives can be captured: update ... gradients | all-reduce | apply as ONE
                # graph, one launch per step (two graphs with an eager collective between them
                # otherwise: gloo, or a runtime that refuses the capture)
                try:
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, pool=pool, stream=side):
                        grads = self._sequence(accumulators, world)
                        torch.distributed.all_reduce(self._flat, op=torch.distributed.ReduceOp.SUM)
                        self._after_reduce(grads, accumulators)
                    ga, self.collective_in_graph = g1, True
                except Exception as e:       # noqa: BLE001  (fall back to the split form)
                    self.collective_capture_error = e
                    torch.cuda.synchronize(dev)
                    ga = None
            if ga is None:
                self.collective_in_graph = False
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, pool=pool, stream=side):
                    grads = self._sequence(accumulators, world)
                if world > 1:
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, pool=pool, stream=side):
                        self._after_reduce(grads, accumulators)
            self._graphs = (sig, ga, gb, grads)
        except Exception as e:
            self.capture_error = e
            self._graphs = None
            torch.cuda.synchronize(dev)
        finally:
            if gc_was_on:
                gc.enable()
        self._republish()
        return err_now
