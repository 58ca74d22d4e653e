#!/usr/bin/env python3
"""Compact per-kernel resource table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scratch/kres.py tfrt_trace3d.hip [name-filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tensorflowraytrace_amd import _build
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = [_build._hipcc(), *_build.HIPCC_FLAGS, *extra, "-I", _build.INCLUDE, "-I", _build.CSRC, "-c",
       os.path.join(_build.CSRC, src), "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?)(?: \[-Rpass)", line) or re.search(r"remark: +(.*?)(?: \[-Rpass)", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if not rows: sys.exit("no kernels found (compile error?)\n" + out[-2000:])
dem = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE).stdout.decode().splitlines()
print(f"{'kernel':58s} {'SGPR':>4s} {'VGPR':>4s} {'AGPR':>4s} {'vspill':>6s} {'sspill':>6s} {'scr':>4s} {'occ':>3s} {'LDS':>6s}")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void tfrt::", "").replace("tfrt::", "")
    if filt and filt not in d: continue
    print(f"{d[:58]:58s} {r.get('TotalSGPRs','?'):>4s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('VGPR Spill', r.get('VGPRs Spill','?')):>6s} {r.get('SGPR Spill', r.get('SGPRs Spill','?')):>6s} {r.get('ScratchSize [bytes/lane]','?'):>4s} {r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>6s}")
