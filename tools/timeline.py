"""Kernel timeline of the last iteration from a rocprofv3 --kernel-trace csv: start offset, duration,
gap to the previous kernel, name.  python tools/timeline.py DIR [N_last_kernels]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(f'{sys.argv[1]}/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}')
    prev_end = e
