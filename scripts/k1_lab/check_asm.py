#!/usr/bin/env python3
"""Audit a K1 .s: no compiler-generated AGPR traffic outside inline-asm statements, no scratch
access (spill traffic) inside the step loops (loop depth >= 2; the chained kernel's item loop, depth
1, may keep a few lane constants in scratch between items), no VALU write of an MFMA operand right in front of an MFMA, M0 only inside the LDS-DMA statements,
no VALU-written SGPR (v_readlane / v_readfirstlane: an SGPR spill reload) read as the descriptor or
scalar offset of an LDS-DMA load within the 5 wait states the hardware needs."""
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
# M0 (the LDS-DMA destination) belongs to the DMA statements: the compiler must not use it, and
# every DMA statement writes it itself right in front of its load
m0bad, dmas = [], 0
inasm = False
for n, l in enumerate(lines, 1):
    if ';;#ASMSTART' in l or '; ;#ASMSTART' in l: inasm = True
    elif ';;#ASMEND' in l or '; ;#ASMEND' in l: inasm = False
    elif re.search(r'\bm0\b', l) and not l.strip().startswith(';') and not inasm:
        m0bad.append((n, l.strip()))
# A request = four 1 KiB pieces behind ONE write of M0: the first piece's statement writes M0 right in
# front of its load (one wait state between: s_nop 0, or the scalar move that carries the tile offset), the
# other three carry the immediate offsets 1024, 2048, 3072 (which move the LDS address and the memory
# address alike) and must follow their first piece in that order with no other write of M0 in between.
prev_off = None
for k, (n, l) in enumerate(code):
    if l.startswith('s_mov_b32 m0,'): prev_off = 'm0'
    if re.search(r'buffer_load.* lds$', l):
        dmas += 1
        mo = re.search(r'offset:(\d+)', l)
        off = int(mo.group(1)) if mo else 0
        if off == 0:
            if k < 2 or not code[k - 2][1].startswith('s_mov_b32 m0,') or not (
                    code[k - 1][1] == 's_nop 0' or re.match(r's_mov_b32 s\d+, s\d+$', code[k - 1][1])):
                m0bad.append((n, l))
        elif off not in (1024, 2048, 3072) or prev_off != off - 1024:
            m0bad.append((n, l))
        prev_off = off
# VALU write of an SGPR -> VMEM read of it needs 5 wait states; hipcc pads its own instructions,
# not the inside of an asm statement (which starts with s_mov m0 + s_nop 0 = 2 states)
def sregs(tok):
    m = re.match(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r's(\d+)$', tok)
    return {int(m.group(1))} if m else set()
shaz = []
for k, (n, l) in enumerate(code):
    if not re.search(r'buffer_load.* lds$', l): continue
    ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
    used = sregs(ops[1]) | sregs(ops[2].split()[0])
    states = 0
    for back in range(1, 8):
        if k - back < 0 or states >= 5: break
        pn, pl = code[k - back]
        if pl.startswith('v_readlane_b32') or pl.startswith('v_readfirstlane_b32'):
            if sregs(pl.split(None, 1)[1].split(',')[0].strip()) & used: shaz.append((n, pl, l))
        m = re.match(r's_nop (\d+)', pl)
        states += (int(m.group(1)) + 1) if m else 1
# scratch traffic by loop depth (hipcc annotates every block label with its loop depth)
deep, depth, per_depth = [], 0, {}
for n, l in enumerate(lines, 1):
    if re.match(r'\.LBB\d+_\d+:', l):
        m = re.search(r'Depth=(\d+)', l)
        depth = int(m.group(1)) if m else 0
    elif re.match(r'\S+:\s*(;.*)?$', l) and not l.startswith('.L'):
        depth = 0  # a new function
    if re.match(r'\s*scratch_(load|store)', l):
        per_depth[depth] = per_depth.get(depth, 0) + 1
        if depth >= 2: deep.append((n, l.strip()))
print("scratch accesses by loop depth:", dict(sorted(per_depth.items())), " inside step loops:", len(deep))
for h in deep[:6]: print(h)
print("VALU-written SGPR -> LDS-DMA descriptor within 5 wait states:", len(shaz))
for h in shaz[:6]: print(h)
print("LDS-DMA loads:", dmas, " M0 uses outside / malformed DMA statements:", len(m0bad))
for n, l in m0bad[:6]: print(n, l)
print("VALU write -> MFMA operand within 2 instructions:", len(haz))
for h in haz[:6]: print(h)
print("compiler uses of a[0:223] outside asm:", len(bad))
for n, l in bad[:10]: print(n, l)
# vector-register spills per kernel (the metadata lists .name before .vgpr_spill_count? no fixed
# order: collect both per kernel entry)
entry = {}
for line in open(sys.argv[1]):
    t = line.strip()
    if t.startswith('- .agpr_count') or t.startswith('- .args'):
        entry = {}
    m = re.match(r'\.?-? ?\.(name|vgpr_spill_count):\s+(\S+)', t)
    if m:
        entry[m.group(1)] = m.group(2)
        if len(entry) == 2:
            print("spills %s %s" % (entry['vgpr_spill_count'], entry['name']))
            entry = {}
for key in ('.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size', '.vgpr_count', '.agpr_count', '.sgpr_count'):
    for line in open(sys.argv[1]):
        if key in line: print(line.strip())
sys.exit(1 if (bad or haz or m0bad or shaz or deep) else 0)
