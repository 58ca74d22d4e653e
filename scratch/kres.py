"""Register / LDS / scratch use of the kernels in a hipcc -S listing.  Usage: kres.py FILE.s [filter]"""
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^([_A-Za-z0-9]+):\s*;\s*@\1\n', t, re.M):
    name = m.group(1)
    if flt not in name: continue
    seg = t[m.end():]
    e = seg.find('.end_amdhsa_kernel')
    if e < 0: continue
    k = seg[e:e + 2500]
    def g(key):
        r = re.search(r';\s*' + key + r':\s*(\d+)', k)
        return r.group(1) if r else '?'
    nins = sum(1 for l in seg[:e].split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';')))
    print(f"{name[:70]:70s} vgpr {g('NumVgprs'):>4s} sgpr {g('NumSgprs'):>4s} scratch {g('ScratchSize'):>4s} lds {g('LDSByteSize'):>6s} occ {g('Occupancy'):>2s} instr {nins}")
