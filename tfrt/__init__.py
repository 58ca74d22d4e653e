"""
``tfrt`` import shim: exposes the MI355X-native implementation under the reference's package
and module names (``tfrt.engine``, ``tfrt.boundaries``, ``tfrt.sources``, ...), so scripts
written against ecpoppenheimer/TensorFlowRayTrace import the same names.
"""
import importlib
import sys

_MODULES = ("engine", "boundaries", "sources", "distributions", "operation", "materials",
            "optimizer", "update", "geometry", "mesh_tools", "drawing", "analyze")
for _m in _MODULES:
    _mod = importlib.import_module("tensorflowraytrace_amd." + _m)
    sys.modules[__name__ + "." + _m] = _mod
    globals()[_m] = _mod
