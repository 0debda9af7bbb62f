"""Highest VGPR index touched per basic block of one kernel's assembly (a crude map of where the register
pressure of a build sits):   python tools/asm_pressure.py kernel.s [min]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
blk, mx, n, first = 'entry', -1, 0, 1
out = []
for i, ln in enumerate(lines, 1):
    m = re.match(r'^(\.LBB\S+):\s*(;.*)?', ln)
    if m:
        out.append((first, blk, mx, n))
        blk, mx, n, first = m.group(1) + ' ' + (m.group(2) or '')[:90], -1, 0, i
        continue
    code = ln.split(';')[0]
    if not code.strip() or code.strip().startswith('.'):
        continue
    n += 1
    for a, b in re.findall(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', code) and [(x[0], x[2]) for x in re.findall(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', code)]:
        v = int(a) if a else int(b)
        mx = max(mx, v)
out.append((first, blk, mx, n))
for first, blk, mx, n in out:
    if mx >= thr:
        print(f'{first:6d} {n:5d} instr  max v{mx:<4d} {blk}')
