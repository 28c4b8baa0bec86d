#!/usr/bin/env python3
"""Audit a K1 .s: no compiler-generated AGPR traffic outside inline-asm statements, no spills."""
import re, sys
inasm = False
bad = []
for n, line in enumerate(open(sys.argv[1]), 1):
    if ';;#ASMSTART' in line or '; ;#ASMSTART' in line: inasm = True
    elif ';;#ASMEND' in line or '; ;#ASMEND' in line: inasm = False
    elif not inasm and not line.strip().startswith(';') and not line.strip().startswith('.'):
        # the kernel owns a[0:191] by literal name; the compiler may use a192 and up
        for m in re.finditer(r'\ba\[?(\d+)', line):
            if int(m.group(1)) < 192:
                bad.append((n, line.rstrip()))
                break
print("compiler uses of a[0:191] outside asm:", len(bad))
for n, l in bad[:10]: print(n, l)
for key in ('.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size', '.vgpr_count', '.agpr_count', '.sgpr_count'):
    for line in open(sys.argv[1]):
        if key in line: print(line.strip())
sys.exit(1 if bad else 0)
