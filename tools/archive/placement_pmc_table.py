"""Per-dispatch table (duration from the kernel trace + counters) of one or more rocprofv3 --pmc
passes over tools/placement_pmc_run.py.   python tools/placement_pmc_table.py OUT.json DIR [DIR...]"""
import csv
import glob
import json
import sys
import collections

out = {}
for d in sys.argv[2:]:
    dur, ctr, name = {}, collections.defaultdict(dict), {}
    for f in glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            did = int(r['Dispatch_Id'])
            dur[did] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
            name[did] = r['Kernel_Name']
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            did = int(r['Dispatch_Id'])
            ctr[did][r['Counter_Name']] = ctr[did].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            name.setdefault(did, r['Kernel_Name'])
    rows = []
    for did in sorted(ctr):
        k = name[did]
        short = 'K1h' if 'k_bloch_fwd_lines' in k else 'K3' if 'k_bloch_bwd_lines' in k else \
            'K0' if 'k_rfgr2beff<' in k else None
        if short:
            rows.append(dict(kernel=short, dispatch=did, ms=round(dur.get(did, float('nan')), 4), **ctr[did]))
    out[d.rstrip('/').split('/')[-1]] = rows
    print('==', d)
    for kname in ('K0', 'K1h', 'K3'):
        rs = [r for r in rows if r['kernel'] == kname]
        if not rs:
            continue
        keys = [k_ for k_ in rs[0] if k_ not in ('kernel', 'dispatch', 'ms')]
        print(kname, 'ms | ' + ' | '.join(keys))
        for r in sorted(rs, key=lambda r: r['ms']):
            print(f"   {r['ms']:8.3f} | " + ' | '.join(f'{r[k_]:.4g}' for k_ in keys))
json.dump(out, open(sys.argv[1], 'w'), indent=1)
