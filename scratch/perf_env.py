"""Step time at a ray count for several values of an environment knob.  perf_env.py RAYS VAR v1 v2 ..."""
import sys, os, subprocess
rays, var, vals = sys.argv[1], sys.argv[2], sys.argv[3:]
here = os.path.dirname(os.path.abspath(__file__))
for v in vals:
    env = dict(os.environ); 
    if v != "default": env[var] = v
    out = subprocess.run([sys.executable, os.path.join(here, "prof_step.py"), rays, "fused", "50"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(f"{var}={v}: {out[-1] if out else '?'}", flush=True)
