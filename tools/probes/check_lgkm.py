#!/usr/bin/env python3
"""Development probe: walk a gfx950 kernel's ISA (hipcc -S) and flag any instruction that touches a VGPR which is the destination of an
LDS operation still in flight according to the s_waitcnt lgkmcnt bookkeeping (LDS operations return in order; SMEM does not, so a block
with scalar loads in flight only counts as drained by lgkmcnt(0)).  usage: check_lgkm.py kernel.s [mangled-name-prefix]"""
import re, sys
path, prefix = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
lines = [l.rstrip() for l in open(path)]
if prefix:
    on, sel = False, []
    for l in lines:
        if l.startswith(prefix): on = True
        if on and ".Lfunc_end" in l: on = False
        if on: sel.append(l)
    lines = sel
def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
inflight = []   # (dest regs, text, lineno), in issue order
smem = 0
n_flag = 0
for ln, l in enumerate(lines):
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): 
        if t.startswith(".LBB"): pass
        continue
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            if n == 0: smem = 0
            if smem == 0:
                while len(inflight) > n: inflight.pop(0)
        continue
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        smem += 1
        continue
    touched = set()
    for o in ops: touched |= regs(o.split(" ")[0])
    for d, txt, l0 in inflight:
        if d & touched:
            n_flag += 1
            if n_flag <= 12: print(f"line {ln}: `{t[:80]}` touches {sorted(d & touched)} = destination of in-flight (line {l0}) `{txt[:70]}`")
    if op.startswith("ds_read") or op.startswith("ds_swizzle") or op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        inflight.append((regs(ops[0]), t, ln))
    elif op.startswith("ds_write") or op.startswith("ds_"):
        inflight.append((set(), t, ln))
print("flagged", n_flag, "of", len(lines), "lines")
