"""Print the average duration of our kernels from a rocprofv3 ``--kernel-trace --stats
--output-format csv`` output directory.  ``python tools/kstats.py DIR [label]``."""
import csv
import glob
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
for f in glob.glob(f'{d}/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Name']
        if 'k_' not in name or 'at::' in name:
            continue
        short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:90]
        print(f"{label:>8}  {float(r['AverageNs']) / 1e6:9.3f} ms  x{r['Calls']:>4}  {short}")
