"""Instruction mix per basic block of one kernel's assembly:  python tools/asm_mix.py kernel.s [min VALU per block]"""
import collections
import re
import sys
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
blk = 'entry'; order = [blk]; c = collections.defaultdict(collections.Counter)
for ln in open(sys.argv[1]):
    m = re.match(r'^(\.LBB\S+):', ln)
    if m:
        blk = m.group(1); order.append(blk); continue
    op = ln.split(';')[0].split()
    if not op:
        continue
    o = op[0]
    for key in ('ds_read', 'ds_write', 's_barrier', 'global_load', 'global_store', 'scratch', 'v_fma_f64', 's_waitcnt', 's_load'):
        if o.startswith(key):
            c[blk][key] += 1
    if o.startswith('v_'):
        c[blk]['valu'] += 1
for b in order:
    if c[b]['valu'] >= thr:
        print(b, dict(c[b]))
