"""
Gradient-descent driver (tfrt/optimizer.py:8-442) on torch autograd + the HIP reverse sweep.

One step = ``system.update()`` (constraints, parameters -> vertices -> faces) ->
``engine.ray_trace(trace_depth)`` -> user ``error_function(engine, ...)`` -> gradient ->
[all-reduce over ray shards] -> non-finite -> 0, scale, clip, accumulator matmul
(optimizer.py:223-257) -> SGD apply (optimizer.py:316) -> smoothing (optimizer.py:261-282).

Effective update rule of the reference: it builds ``tf.optimizers.SGD(nesterov=True)`` with
the Keras defaults (learning rate 0.01, momentum 0.0) and later only *assigns* ``momentum``
(optimizer.py:103,128-132); in Keras OptimizerV2 the momentum branch is chosen at
construction, so the step is plain ``p -= 0.01 * processed_grad``.  That is the default here
(``sgd_learning_rate=0.01``, ``apply_momentum=False``); ``apply_momentum=True`` enables the
Nesterov rule the constructor presumably intended.
"""
import time
import weakref

import numpy as np
import torch

from . import distributed as tdist
from . import ops
from .fused_step import FusedStep, GoalError, RowwiseError, _NotInPlace  # noqa: F401  (GoalError, RowwiseError: public API)


class DeferredScalar:
    """The step error as the reference returns it (``error.numpy()``, optimizer.py:320), except
    that the device->host read happens when the value is first used (``float(e)``, printing,
    comparing, arithmetic) instead of inside ``single_step``: a blocking read there would drain
    the GPU queue once per step and leave the device idle while the host prepares the next
    trace."""
    __slots__ = ("_tensor", "_value", "__weakref__")

    def __init__(self, tensor):
        self._tensor = tensor.detach()
        self._value = None

    def __float__(self):
        if self._value is None:
            self._value = float(self._tensor)
            self._tensor = None
        return self._value

    def item(self):
        return float(self)

    def numpy(self):
        return np.float64(float(self))

    def __array__(self, dtype=None, copy=None):
        return np.asarray(float(self), dtype=dtype)

    def __repr__(self):
        return repr(float(self))

    __str__ = __repr__

    def __format__(self, spec):
        return format(float(self), spec)

    def __bool__(self):
        return bool(float(self))

    def __hash__(self):
        return hash(float(self))

    def __eq__(self, other):
        return float(self) == other

    def __lt__(self, other):
        return float(self) < other

    def __le__(self, other):
        return float(self) <= other

    def __gt__(self, other):
        return float(self) > other

    def __ge__(self, other):
        return float(self) >= other

    def __neg__(self):
        return -float(self)

    def __abs__(self):
        return abs(float(self))

    def __add__(self, other):
        return float(self) + other

    __radd__ = __add__

    def __sub__(self, other):
        return float(self) - other

    def __rsub__(self, other):
        return other - float(self)

    def __mul__(self, other):
        return float(self) * other

    __rmul__ = __mul__

    def __truediv__(self, other):
        return float(self) / other

    def __rtruediv__(self, other):
        return other / float(self)

    def __pow__(self, other):
        return float(self) ** other


class SGD_Optimizer:
    def __init__(self, engine, parameters, error_function, trace_depth, momentum=0.0,
                 learning_rate=1.0, individual_lr=None, grad_clip="default", clip_mode="common",
                 clip_scale=10.0, sgd_learning_rate=0.01, apply_momentum=False, speculative=False,
                 fused="auto", graph="auto"):
        self.engine = engine
        if type(parameters) is list or type(parameters) is tuple:
            self.parameters = parameters
        else:
            raise ValueError("SGD_Optimizer: parameters must be a list of tf.variable")
        self.error_function = error_function
        self.trace_depth = trace_depth
        self.sgd_learning_rate = sgd_learning_rate
        self.apply_momentum = apply_momentum
        # ``speculative=True`` (opt-in) overlaps host and device on the generic path: the per-class
        # ray counts of the trace are guessed from the previous step and verified after the
        # gradient has been enqueued.  On a wrong guess the error function has ALREADY run once on
        # ray sets cut with the stale counts (rows beyond the true count are uninitialised) and is
        # run again -- only for pure error functions; the default reads the counts first.
        self.speculative = speculative
        self.speculation_misses = 0
        # start pessimistic: the first steps read the ray counts (blocking) and speculation
        # begins once they have repeated (5 equal steps), instead of paying for wrong guesses
        self._miss_rate = 1.0
        self._last_counts = None
        self._velocity = [None] * len(self.parameters)
        self.momentum = momentum
        self.learning_rate = learning_rate
        self.individual_lr = individual_lr
        self.clip_scale = clip_scale
        self.grad_clip = self.clip_scale * learning_rate if grad_clip == "default" else grad_clip
        self.clip_mode = clip_mode
        self.suppress_warnings = False
        self.iterations = 0
        self.last_error_terms = 0
        self._acc_cache = {}
        # ``fused``: run a step whose error function is a ``GoalError`` as one fixed launch
        # sequence (fused_step.FusedStep: no host read of the ray counts, built-in error and seed
        # kernel); ``graph``: replay that sequence from a captured HIP graph after a few steps.
        # "auto" = whenever possible; False = always the generic path.
        self.fused = fused
        self.graph = graph
        self._fused_step = None
        self._pending_error = None

    @property
    def momentum(self):
        return self._momentum

    @momentum.setter
    def momentum(self, val):
        if 0.0 <= val <= 1.0:
            self._momentum = val
        else:
            raise ValueError("SGD_Optimizer: Momentum must be between 0 and 1.")

    @property
    def individual_lr(self):
        return self._individual_lr

    @individual_lr.setter
    def individual_lr(self, val):
        if val is None:
            self._individual_lr = [1.0] * len(self.parameters)
            return
        try:
            if len(val) != len(self.parameters):
                raise ValueError(
                    "SGD_Optimizer: individual_lr must have as many elements as there are "
                    "parameters.")
        except TypeError as e:
            raise TypeError(
                "SGD_Optimizer: individual_lr must have as many elements as there are "
                "parameters.") from e
        self._individual_lr = val

    def convert_to_plist(self, data):
        p_count = len(self.parameters)
        if type(data) is list or type(data) is tuple:
            if len(data) == p_count:
                return data
            raise ValueError("SGD_Optimizer: plist arguments must have one element per parameter.")
        return [data] * p_count

    def convert_to_lrlist(self, lr, steps):
        try:
            return np.linspace(lr[0], lr[1], steps)
        except TypeError:
            return [lr] * steps

    # ------------------------------------------------------------------------ gradient
    def raw_gradient(self, *args, **kwargs):
        """update -> trace -> error -> gradient, summed over ray shards.  Returns
        (grads, error_sum, n_error_terms)."""
        self.engine.clear_ray_history()
        self.engine.optical_system.update()
        # speculate only while guesses are mostly right (with many rays some ray changes class
        # almost every step; a blocking count read is then cheaper than re-evaluating)
        speculate = self.speculative and self._miss_rate < 0.4
        if hasattr(self.engine, "speculative_counts"):
            self.engine.speculative_counts = speculate
        self.engine.ray_trace(self.trace_depth)

        def evaluate(retain=False):
            error = self.error_function(self.engine, *args, **kwargs)
            error_sum = error.sum()
            if error_sum.requires_grad:
                # the reverse sweep is ~25 tiny launches: run it on this thread instead of
                # handing every node to autograd's device thread (~0.1 ms per step)
                with torch.autograd.set_multithreading_enabled(False):
                    g = torch.autograd.grad(error_sum, self.parameters, allow_unused=True,
                                            retain_graph=retain)
            else:
                g = [None] * len(self.parameters)
            return g, error_sum, error.numel()

        spec = speculate and hasattr(self.engine, "verify_trace") and \
            "pending" in (self.engine.last_trace or {})
        grads, error_sum, n_terms = evaluate(retain=spec)
        if spec:
            ok = self.engine.verify_trace()
            self._miss_rate = 0.8 * self._miss_rate + (0.0 if ok else 0.2)
            if not ok:  # ray counts changed since the last step
                self.speculation_misses += 1
                grads, error_sum, n_terms = evaluate()
        elif self.speculative:
            # not speculating: compare the counts with the previous step's to keep the estimate
            prev, cur = self._last_counts, (self.engine.last_trace or {}).get("raw_counts")
            if prev is not None and cur is not None and prev.shape == cur.shape:
                same = bool((prev == cur).all())
                self._miss_rate = 0.8 * self._miss_rate + (0.0 if same else 0.2)
        if self.speculative and self.engine.last_trace is not None:
            self._last_counts = self.engine.last_trace.get("raw_counts")
        if hasattr(self.engine, "speculative_counts"):
            self.engine.speculative_counts = False
        fixed = []
        for g, p in zip(grads, self.parameters):
            if g is None:
                if not self.suppress_warnings:
                    print(
                        "Warning: SGD_Optimizer.process_gradient encountered a possible issue:  "
                        "The gradient was likely None, which can mean that the error does not "
                        "depend on it.  The gradient will be set to zero and future instances "
                        "of this message will be suppressed.")
                    self.suppress_warnings = True
                g = torch.zeros_like(p)
            fixed.append(g)
        fixed, error_sum, n_terms = tdist.all_reduce_step(fixed, error_sum.detach(), n_terms)
        return fixed, error_sum, n_terms

    def _matrix_product(self, cache, key, matrix, vec):
        """``matrix @ vec`` for an accumulator / smoother: the matrix is converted to CSR once
        (cached on the object it came from) and multiplied by the tfrt_csr_matvec kernel."""
        entry = cache.get(key)
        if entry is None or entry[0] is not matrix:
            entry = (matrix, ops.CsrMatrix(matrix, vec.device))
            cache[key] = entry
        return entry[1].matvec(vec)

    def process_gradient(self, accumulators, *args, lr_scale=1.0, **kwargs):
        """optimizer.py:187-258.  Returns (processed grads, mean error)."""
        processed, mean, _ = self._process(accumulators, args, kwargs, lr_scale, apply=False)
        return processed, mean

    def _process(self, accumulators, args, kwargs, lr_scale, apply):
        """Gradient processing; with ``apply`` the SGD update of every parameter that has no
        accumulator is fused into the processing kernel (returns which ones were applied)."""
        grads, error_sum, n_terms = self.raw_gradient(*args, **kwargs)
        self.last_error_terms = n_terms
        processed, applied = [], []
        plain_sgd = not (self.apply_momentum and self._momentum > 0.0)
        for i, grad in enumerate(grads):
            scale = lr_scale * self.individual_lr[i] * self.learning_rate
            if self.clip_mode == "common":
                clp = self.grad_clip
            else:
                clp = self.individual_lr[i] * self.clip_scale * self.learning_rate * lr_scale
            p = self.parameters[i]
            fuse = (apply and plain_sgd and accumulators[i] is None and p.is_contiguous()
                    and p.dtype == grad.dtype and p.shape == grad.shape)
            with torch.no_grad():
                grad = ops.sgd_process(grad, scale, clp, param=p if fuse else None,
                                       sgd_learning_rate=self.sgd_learning_rate)
            applied.append(fuse)
            if accumulators[i] is not None:
                grad = self._matrix_product(self._acc_cache, i, accumulators[i], grad)
            processed.append(grad)
        # (tf.reduce_mean, optimizer.py:257: the mean of no error terms is NaN)
        if isinstance(n_terms, torch.Tensor):
            mean = torch.where(n_terms > 0, error_sum / torch.clamp(n_terms, min=1.0),
                               torch.full_like(error_sum, float("nan")))
        else:
            mean = error_sum / n_terms if n_terms > 0 else error_sum * float("nan")
        return processed, mean, applied

    _smoother_cache = {}

    @staticmethod
    def smooth(parameters, smoother):
        """optimizer.py:261-282: ``parameters <- smoother @ parameters`` in place."""
        if smoother is not None:
            with torch.no_grad():
                cache = SGD_Optimizer._smoother_cache
                entry = cache.get(id(smoother))
                if entry is None or entry[0] is not smoother:
                    if len(cache) > 64:
                        cache.clear()
                    entry = (smoother, ops.CsrMatrix(smoother, parameters.device))
                    cache[id(smoother)] = entry
                parameters.copy_(entry[1].matvec(parameters.detach()))

    def apply_gradients(self, grads, skip=None):
        with torch.no_grad():
            for i, (g, p) in enumerate(zip(grads, self.parameters)):
                if skip is not None and skip[i]:
                    continue  # already applied by the fused processing kernel
                lr = self.sgd_learning_rate
                if self.apply_momentum and self._momentum > 0.0:
                    v = self._velocity[i]
                    if v is None:
                        v = torch.zeros_like(p)
                    v = self._momentum * v - lr * g
                    self._velocity[i] = v
                    p.add_(self._momentum * v - lr * g)  # Nesterov form used by Keras
                else:
                    p.add_(g, alpha=-lr)

    def single_step(self, accumulators, *args, lr_scale=1.0, momentum=0.0, verbose=False,
                    **kwargs):
        """optimizer.py:284-320."""
        self.momentum = momentum
        accumulators = self.convert_to_plist(accumulators)
        if self.fused in ("auto", True) and FusedStep.eligible(self, args, kwargs):
            if self._fused_step is None:
                self._fused_step = FusedStep(self, graph=self.graph)
            # the step writes its error into a buffer the next step overwrites: an error value
            # nobody has read yet is copied out just before that (and only then: no copy launch
            # per step when the caller drops or reads the value right away)
            prev = self._pending_error() if self._pending_error is not None else None
            if prev is not None and prev._value is None and prev._tensor is not None:
                prev._tensor = prev._tensor.clone()
            try:
                err3 = self._fused_step.step(accumulators, lr_scale)   # {sum, n_terms, mean}, device
            except _NotInPlace:
                # (a RowwiseError whose source is not traced in place after all: the generic path,
                # from now on -- FusedStep.rowwise_ready is off for this engine)
                return self.single_step(accumulators, *args, lr_scale=lr_scale, momentum=momentum,
                                        verbose=verbose, **kwargs)
            self.iterations += 1
            self.last_error_terms = err3[1]
            err = DeferredScalar(err3[2])
            self._pending_error = weakref.ref(err)
            if verbose:
                print(f"step {self.iterations} error: {err}")
            return err
        grads, error, applied = self._process(accumulators, args, kwargs, lr_scale, apply=True)
        self.apply_gradients(grads, skip=applied)
        self.iterations += 1
        err = DeferredScalar(error) if isinstance(error, torch.Tensor) and error.is_cuda \
            else float(error)
        if verbose:
            print(f"step {self.iterations} error: {err}")
        return err

    def training_routine(self, routine, post_step=None, report_frequency=1, show_time=True):
        """optimizer.py:322-442: list of phase dicts, each updating the running phase."""
        phase = {"steps": 10, "learning_rate": 1.0, "momentum": 0.0, "accumulators": None,
                 "smoothers": None, "erf_args": [], "erf_kwargs": {}, "individual_lr": None}
        self.iterations = 0
        phase_count = len(routine)
        total_iterations = 0
        steps = phase["steps"]
        start_time = time.time()
        for new_phase in routine:
            steps = new_phase.get("steps", steps)
            total_iterations += steps
        current_phase = 0
        is_rank0 = tdist.rank() == 0
        for new_phase in routine:
            current_phase += 1
            phase_iterations = 0
            phase.update(new_phase)
            phase["accumulators"] = self.convert_to_plist(phase["accumulators"])
            phase["smoothers"] = self.convert_to_plist(phase["smoothers"])
            lrs = self.convert_to_lrlist(phase["learning_rate"], phase["steps"])
            self.individual_lr = phase["individual_lr"]
            for i in range(phase["steps"]):
                error = self.single_step(phase["accumulators"], *phase["erf_args"],
                                         lr_scale=lrs[i], momentum=phase["momentum"],
                                         verbose=False, **phase["erf_kwargs"])
                for p, s in zip(self.parameters, phase["smoothers"]):
                    self.smooth(p, s)
                phase_iterations += 1
                if report_frequency != 0 and is_rank0 and self.iterations % report_frequency == 0:
                    print(f"Phase {current_phase}/{phase_count}, "
                          f"step {phase_iterations}/{phase['steps']}, "
                          f"total {self.iterations}/{total_iterations}-"
                          f"{100 * self.iterations / total_iterations:.1f}%.  Error: {error}.")
                if post_step:
                    post_step()
        total_time = time.time() - start_time
        if show_time and is_rank0:
            print(f"Completed training routine.  Took {total_time} seconds.")
            print(f"Steps took an average of {total_time / max(total_iterations, 1)} seconds per step.")
