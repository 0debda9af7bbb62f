"""Summarise a rocprofv3 ``--pmc ... --kernel-trace --output-format csv`` directory: per kernel,
mean counter values per dispatch.  ``python tools/pmc_sq.py DIR``"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{sys.argv[1]}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        k = k.split('(')[0]
        if not k.startswith('k_'):
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k[:80])
    for c, v in sorted(cs.items()):
        print(f'    {c:28s} {sum(v) / len(v):.5g}  (n={len(v)})')
