"""
In-tree build of libtfrt_hip.so (hipcc, gfx950 only).

``python -m tensorflowraytrace_amd._build`` or ``build()`` compiles every ``csrc/*.hip`` to an
object and links ``tensorflowraytrace_amd/libtfrt_hip.so``.  hipcc cross-compiles for gfx950
without a GPU present.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libtfrt_hip.so")
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-munsafe-fp-atomics",     # float64 atomicAdd -> global_atomic_add_f64, no CAS loop
    "-fno-gpu-rdc",
    "-fno-slp-vectorize",      # keep scalar v_fma_f32 in the filter loop (no v_pk_fma_f32)
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, extra_flags=(), drop_flags=(), lib_path=None, tag=""):
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "tfrt_hip.h"))
    objdir = os.path.join(CSRC, "_obj" + tag)
    lib_out = lib_path or LIB
    flags = [f for f in HIPCC_FLAGS if f not in drop_flags]
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    for src in sources:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src[:-4] + ".o")
        objs.append(op)
        if (not force and os.path.exists(op)
                and os.path.getmtime(op) >= _newest([sp] + headers)):
            continue
        cmd = [hipcc, *flags, *extra_flags, "-I", INCLUDE, "-I", CSRC, "-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    if force or procs or not os.path.exists(lib_out) or os.path.getmtime(lib_out) < _newest(objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_out, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return lib_out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
