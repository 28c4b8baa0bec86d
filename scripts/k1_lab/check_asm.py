#!/usr/bin/env python3
"""Audit a K1 .s: no compiler-generated AGPR traffic outside inline-asm statements, no spills."""
import re, sys
inasm = False
bad = []
for n, line in enumerate(open(sys.argv[1]), 1):
    if ';;#ASMSTART' in line or '; ;#ASMSTART' in line: inasm = True
    elif ';;#ASMEND' in line or '; ;#ASMEND' in line: inasm = False
    elif not inasm and not line.strip().startswith(';') and not line.strip().startswith('.'):
        # the kernel owns a[0:223] by literal name; the compiler may use a224 and up
        for m in re.finditer(r'\ba\[?(\d+)', line):
            if int(m.group(1)) < 224:
                bad.append((n, line.rstrip()))
                break
# VALU write of an MFMA A/B operand within the 2 instructions in front of the MFMA statement
# (the compiler does not know the statement is an MFMA and pads nothing)
lines = [l.rstrip() for l in open(sys.argv[1])]
code = [(n, l.strip()) for n, l in enumerate(lines, 1)
        if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
haz = []
for k, (n, l) in enumerate(code):
    if not l.startswith('v_mfma'): continue
    ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
    src = regs(ops[1]) | regs(ops[2])
    for back in (1, 2):
        if k - back < 0: break
        pn, pl = code[k - back]
        if pl.startswith('v_mfma') or not pl.startswith('v_'): continue
        dst = regs(pl.split(None, 1)[1].split(',')[0].strip())
        if dst & src: haz.append((n, pl, l))
print("VALU write -> MFMA operand within 2 instructions:", len(haz))
for h in haz[:6]: print(h)
print("compiler uses of a[0:223] outside asm:", len(bad))
for n, l in bad[:10]: print(n, l)
for key in ('.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size', '.vgpr_count', '.agpr_count', '.sgpr_count'):
    for line in open(sys.argv[1]):
        if key in line: print(line.strip())
sys.exit(1 if (bad or haz) else 0)
