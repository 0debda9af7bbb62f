"""HBM bytes per launch of every product kernel from two rocprofv3 PMC passes (FETCH_SIZE in one
run, WRITE_SIZE in another -- never together, never with tracing domains besides --kernel-trace):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -- python3 bench.py ...
    python tools/pmc_bytes.py F W [algorithmic.json]

Units and corrections as MI355X_MICROARCH.md prescribes: counters are KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads, so it is doubled.  (4-B-per-lane
accesses -- the SoA history stream -- are "uncalibrated" per the guide; the doubled figure is
checked against the known byte count of that stream in DESIGN.md.)  Prints JSON: kernel ->
{fetch_bytes, write_bytes, total, launches}; with `algorithmic.json` (kernel substring -> bytes) the
ratio traffic / algorithmic is added.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            k = k.split('(')[0]
            if k.startswith('k_'):
                acc[k].append(float(r['Counter_Value']))
    return acc


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
alg = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2 if fetch.get(k) else None
    w = sum(write[k]) / len(write[k]) * 1024 if write.get(k) else None
    e = {'fetch_bytes_x2': f, 'write_bytes': w, 'total': (f or 0) + (w or 0),
         'launches': len(fetch.get(k) or write.get(k))}
    for sub, b in alg.items():
        if sub in k:
            e['algorithmic'] = b
            e['traffic_over_algorithmic'] = e['total'] / b
    out[k] = e
print(json.dumps(out, indent=1))
