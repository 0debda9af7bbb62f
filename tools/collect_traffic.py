"""Turn two rocprofv3 PMC passes over `bench.py --no-cpu --no-fused` (FETCH_SIZE in one run,
WRITE_SIZE in another -- never together with tracing domains) into profiles/traffic.json and the
per-dispatch table profiles/r01_bench_n128_nT4096_pmc_hbm.csv.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu --no-fused
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu --no-fused
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write

Units and corrections as MI355X_MICROARCH.md prescribes: the counters are in KiB; on gfx950
FETCH_SIZE counts half of the bytes of 16-B/lane coalesced reads, so it is doubled.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def rows(d, counter):
    out = []
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = r['Kernel_Name']
            name = ('k_bloch_fwd_lines' if 'k_bloch_fwd_lines' in k else
                    'k_rfgr2beff' if 'k_rfgr2beff<' in k else None)
            if name:
                out.append((name, float(r['Counter_Value']), r['Dispatch_Id']))
    return out


fetch, write = rows(sys.argv[1], 'FETCH_SIZE'), rows(sys.argv[2], 'WRITE_SIZE')
mean = lambda xs: sum(xs) / len(xs)  # noqa: E731
kib = 1024.0
res = {}
for name in ('k_bloch_fwd_lines', 'k_rfgr2beff'):
    f = mean([v for n, v, _ in fetch if n == name]) * kib * 2      # gfx950: x2
    w = mean([v for n, v, _ in write if n == name]) * kib
    res[name] = (f, w)
out = {
    'note': 'HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in '
            'separate runs; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 rule for '
            '16 B/lane coalesced reads), workload 128^3 x 4096 fp32',
    'k_bloch_fwd_bytes_per_launch': sum(res['k_bloch_fwd_lines']),
    'k_bloch_fwd_fetch_bytes': res['k_bloch_fwd_lines'][0],
    'k_bloch_fwd_write_bytes': res['k_bloch_fwd_lines'][1],
    'k_rfgr2beff_bytes_per_launch': sum(res['k_rfgr2beff']),
    'k_rfgr2beff_write_bytes': res['k_rfgr2beff'][1],
}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', 'r01_bench_n128_nT4096_pmc_hbm.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['pass', 'Kernel_Name', 'Counter_Name', 'Counter_Value_KiB', 'Dispatch_Id'])
    for n, v, d in fetch:
        w.writerow(['fetch', n, 'FETCH_SIZE', f'{v:.6f}', d])
    for n, v, d in write:
        w.writerow(['write', n, 'WRITE_SIZE', f'{v:.6f}', d])
print(json.dumps(out, indent=1))
