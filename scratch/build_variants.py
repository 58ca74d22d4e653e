"""Builds tuning variants of the library into scratch/variants_live/ (git-ignored, not shipped; it travels to the GPU box: delete it after use).  Usage:
build_variants.py NAME=flags ...   e.g.  tune="-DTFRT_TUNING" """
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tensorflowraytrace_amd import _build
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants_live")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    path = _build.build(extra_flags=tuple(flags.split()), lib_path=os.path.join(out, f"lib_{name}.so"), tag="_" + name)
    print(name, "->", path, flush=True)
