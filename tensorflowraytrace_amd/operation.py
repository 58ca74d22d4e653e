"""
Ray operations (plug-in protocol of tfrt/operation.py:25-333).

In the reference every pass calls each operation's ``main`` in Python.  Here the reaction
(``StandardReaction``: refraction / reflection, operation.py:255-307) is fused into the HIP
pass kernel, so an operation object mainly *declares* what the engine must do: the signature
sets the engine unions (engine.py:1308-1316) and, for ``StandardReaction``, the refractive
index mode.  ``annotate`` / ``preprocess`` / ``postprocess`` hooks are still honoured by
``OpticalEngine.single_pass``.

A user operation that overrides ``main`` (the reference's plug-in point for new reactions) is
run the reference's way: ``ray_trace`` then loops over ``single_pass`` in Python, the projection
(intersection, classification, boundary-data gather) still runs in the kernels, and ``main``
receives the same ``proj_result`` dict and returns ``{"active": {"rays": {...}, "valid": mask}}``.
"""
import torch


class RayOperation:
    """Base class: seven signature properties + four hooks (operation.py:25-162)."""

    def __init__(self, active=True):
        self.active = active

    @property
    def input_signature(self):
        return set()

    @property
    def output_signature(self):
        return set()

    @property
    def optical_signature(self):
        return set()

    @property
    def stop_signature(self):
        return set()

    @property
    def target_signature(self):
        return set()

    @property
    def material_signature(self):
        return set()

    @property
    def simple_ray_inheritance(self):
        return set()

    def annotate(self, engine):
        pass

    def preprocess(self, engine, proj_result):
        pass

    def main(self, engine, proj_result):
        return {}

    def postprocess(self, engine, proj_result, new_rays):
        pass

    @property
    def exclusions(self):
        return set()


class OldestAncestor(RayOperation):
    """Tags every source ray with its index and lets children inherit it
    (operation.py:166-196)."""

    @property
    def input_signature(self):
        return {"oldest_ancestor"}

    @property
    def output_signature(self):
        return self.input_signature

    @property
    def simple_ray_inheritance(self):
        return self.input_signature

    def annotate(self, engine):
        start = 0
        for source in engine.optical_system._sources:
            count = source["x_start"].shape[0]
            source["oldest_ancestor"] = torch.arange(
                start, start + count, device=source["x_start"].device)
            start += count


class StandardReaction(RayOperation):
    """Refraction / reflection at optical boundaries (operation.py:200-307).

    ``refractive_index_type``: ``"index"`` -- boundaries carry ``mat_in`` / ``mat_out``
    indices into ``system.materials`` and rays carry ``wavelength``; ``"value"`` -- boundaries
    carry ``n_in`` / ``n_out`` directly.  The arithmetic itself lives in the HIP pass kernel
    (csrc/trace_math.h ``snell3d`` / ``snell2d_angle``).
    """

    fused = True  # the engine runs this reaction inside the trace kernels

    def __init__(self, refractive_index_type="index", **kwargs):
        super().__init__(**kwargs)
        if refractive_index_type not in {"index", "value"}:
            raise ValueError(
                f"StandardReaction: received invalid value {refractive_index_type}.  "
                "Must be 'index' or 'value'.")
        self._refractive_index_type = refractive_index_type

    @property
    def refractive_index_type(self):
        return self._refractive_index_type

    @property
    def input_signature(self):
        return {"wavelength"} if self._refractive_index_type == "index" else set()

    @property
    def output_signature(self):
        return self.input_signature

    @property
    def simple_ray_inheritance(self):
        return self.input_signature

    @property
    def optical_signature(self):
        if self._refractive_index_type == "index":
            return {"mat_in", "mat_out"}
        return {"n_in", "n_out"}

    @property
    def material_signature(self):
        return {"n"} if self._refractive_index_type == "index" else set()


class GhostThrough(RayOperation):
    """Rays pass straight through optical surfaces (operation.py:311-333).  Implemented with
    the fused reaction by treating every optical boundary as index-matched (n_in = n_out = 1),
    for which Snell's law returns the incoming direction."""

    fused = True
    ghost = True
