"""Step time of the bench workload for several library builds.  perf_variants.py RAYS lib1 lib2 ... ('default' = shipped)"""
import sys, os, subprocess
rays, libs = sys.argv[1], sys.argv[2:]
here = os.path.dirname(os.path.abspath(__file__))
for v in libs:
    env = dict(os.environ)
    if v != "default": env["TFRT_LIB_PATH"] = os.path.join(here, "variants", f"lib_{v}.so")
    out = subprocess.run([sys.executable, os.path.join(here, "prof_step.py"), rays, "fused", "40"], env=env, capture_output=True, text=True)
    lines = out.stdout.strip().splitlines()
    print(f"{v}: {lines[-1] if lines else out.stderr[-300:]}", flush=True)
