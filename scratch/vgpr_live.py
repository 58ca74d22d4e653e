"""VGPR liveness over the ISA of one kernel (hipcc -S output).  vgpr_live.py FILE.s KERNEL_SUBSTRING [LINE]
Prints the pressure profile, the point of highest pressure and, for every register live there (or at
LINE of the extracted kernel), where it was last written and where it is next read."""
import re, sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l))
end = next(i for i in range(start, len(text)) if '.end_amdhsa_kernel' in text[i] or text[i].startswith('.Lfunc_end'))
lines = text[start:end]
def regs(tok):
    out = set()
    for m in re.finditer(r'(?<![\w.])v(\d+)\b', tok): out.add(int(m.group(1)))
    for m in re.finditer(r'(?<![\w.])v\[(\d+):(\d+)\]', tok): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
NODEF = ('global_store', 'ds_write', 'buffer_store', 'flat_store', 'scratch_store', 's_', 'v_cmp_', 'ds_min', 'ds_add', 'ds_max',
         'global_atomic', 'v_readlane', 'v_readfirstlane', 'ds_or', 'ds_and')
ins = []   # (line index, defs, uses, label, branch target, falls through)
label_at = {}
for i, l in enumerate(lines):
    c = l.split(';')[0].rstrip()
    m = re.match(r'^(\.LBB\w+):', c)
    if m: label_at[m.group(1)] = len(ins); continue
    c = c.strip()
    if not c or c.startswith('.') or c.endswith(':'): continue
    parts = c.split(None, 1)
    op = parts[0]; rest = parts[1] if len(parts) > 1 else ''
    tgt = None; fall = True
    if op.startswith('s_cbranch') or op == 's_branch':
        tgt = rest.strip(); fall = op != 's_branch'
    if op == 's_endpgm': fall = False
    ops = rest.split(',')
    if op.startswith(NODEF) and not op.startswith('v_cmpx'):
        d, u = set(), regs(rest)
    else:
        d = regs(ops[0]); u = regs(','.join(ops[1:]))
        if op.startswith(('v_fmac', 'v_mac', 'v_writelane', 'v_pk_fmac', 'v_dot')) or 'dpp' in rest or 'row_' in rest or 'quad_perm' in rest:
            u |= d
    ins.append([i, d, u, tgt, fall])
n = len(ins)
succ = [[] for _ in range(n)]
for k, (i, d, u, tgt, fall) in enumerate(ins):
    if fall and k + 1 < n: succ[k].append(k + 1)
    if tgt in label_at and label_at[tgt] < n: succ[k].append(label_at[tgt])
live_in = [set() for _ in range(n)]; live_out = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for k in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[k]: out |= live_in[s_]
        inn = ins[k][2] | (out - ins[k][1])
        if out != live_out[k] or inn != live_in[k]:
            live_out[k], live_in[k] = out, inn; changed = True
press = [len(x) for x in live_in]
peak = max(range(n), key=lambda k: press[k])
print(f"{n} instructions, peak pressure {press[peak]} at kernel line {ins[peak][0]}: {lines[ins[peak][0]].strip()}")
step = max(1, n // 60)
print("profile:", [max(press[k:k + step]) for k in range(0, n, step)])
at = peak
if len(sys.argv) > 3:
    at = min(range(n), key=lambda k: abs(ins[k][0] - int(sys.argv[3])))
print(f"live at kernel line {ins[at][0]} ({len(live_in[at])}):")
for r in sorted(live_in[at]):
    dl = next((ins[k][0] for k in range(at - 1, -1, -1) if r in ins[k][1]), None)
    ul = next((ins[k][0] for k in range(at, n) if r in ins[k][2]), None)
    print(f"  v{r}: written line {dl}: {lines[dl].strip()[:70] if dl is not None else '-'} | next read line {ul}: {lines[ul].strip()[:60] if ul is not None else '(loop back)'}")
