"""Condense gpurun_out/r03p (tools/runs/r03_profiles.sh) into the tracked profiles/r03_* files:
bench lines, rocprofv3 kernel-stats summaries, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes,
the guide's gfx950 corrections), the K2 instruction mix, the parity ledger.

    python tools/collect_r03.py [gpurun_out/r03p]
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
O = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r03p')
P = os.path.join(ROOT, 'profiles')


def cp(src, dst):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
        print('  ', dst)


def stats(d, dst, keep=('k_',)):
    r"""rocprofv3 --stats: the kernel_stats.csv of the run, our kernels only."""
    fs = glob.glob(os.path.join(O, d, '**', '*kernel_stats.csv'), recursive=True)
    if not fs:
        return
    rows = list(csv.DictReader(open(fs[0])))
    with open(os.path.join(P, dst), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
            if n.startswith(keep):
                w.writerow(r)
    print('  ', dst)


# one collection per directory: the per-run CSVs carry the run id in their names, and a directory that
# gpurun merged two collections into would average (or, before pmc_summary keyed by file, SUM) them
for d in ('pmc_fetch_fwd', 'pmc_write_fwd', 'k2_sq1'):
    fs = glob.glob(os.path.join(O, d, '**', '*counter_collection.csv'), recursive=True)
    assert len(fs) <= 1, f'{d}: {len(fs)} counter files -- remove the local copy of {O} before a new collection'
for src, dst in (('bench_fwd.json', 'r03_bench_n128_nT4096.json'),
                 ('bench_fwd_shard8.json', 'r03_bench_n128_nT4096_shard_of_8.json'),
                 ('bench_fwd_cfg1.json', 'r03_bench_n64_nT1024.json'),
                 ('valu_operand_rates.txt', 'r03_valu_operand_rates.txt'),
                 ('bench_grad_cfg4.json', 'r03_bench_grad_cfg4_n64_nT2048.json'),
                 ('bench_grad128.json', 'r03_bench_grad_n128_nT1024.json'),
                 ('parity_ledger.json', 'r03_parity.json'),
                 ('pytest_gpu_tail.txt', 'r03_pytest_gpu_tail.txt')):
    cp(src, dst)
stats('prof_fwd', 'r03_bench_n128_nT4096_kernel_stats.csv')
stats('prof_grad128', 'r03_grad_n128_nT1024_kernel_stats.csv')
stats('prof_grad_cfg4', 'r03_grad_cfg4_n64_nT2048_kernel_stats.csv')

# PMC traffic: pmc_summary.py merges the FETCH and WRITE passes per workload
py = sys.executable
for label, dirs in (('fwd_128_4096', ('pmc_fetch_fwd', 'pmc_write_fwd')),
                    ('grad_128_1024', ('pmc_fetch_grad128', 'pmc_write_grad128')),
                    ('grad_64_2048', ('pmc_fetch_grad64', 'pmc_write_grad64'))):
    ds = [os.path.join(O, d) for d in dirs if os.path.isdir(os.path.join(O, d))]
    if ds:
        subprocess.run([py, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(O, 'traffic_all.json'),
                        label] + ds, check=True, stdout=subprocess.DEVNULL)
if os.path.exists(os.path.join(O, 'traffic_all.json')):
    T = json.load(open(os.path.join(O, 'traffic_all.json')))
    out = {'note': 'HBM bytes per launch from rocprofv3 PMC passes over tools/run_kernels.py (FETCH_SIZE and '
                   'WRITE_SIZE in separate runs, never with tracing domains; counters are in KiB; on gfx950 '
                   'FETCH_SIZE counts half of the bytes of 16-B/lane coalesced reads, so it is doubled: '
                   'MI355X_MICROARCH.md, HBM); first dispatch of each kernel dropped', 'workloads': {}}
    for label, ks in T.items():
        w = {}
        for k, e in ks.items():
            if 'fetch_bytes_corrected' in e or 'write_bytes' in e:
                w[k] = {'fetch_bytes': e.get('fetch_bytes_corrected'), 'write_bytes': e.get('write_bytes'),
                        'total_bytes': (e.get('fetch_bytes_corrected') or 0) + (e.get('write_bytes') or 0),
                        'FETCH_SIZE_KiB_raw': e.get('FETCH_SIZE'), 'WRITE_SIZE_KiB_raw': e.get('WRITE_SIZE'),
                        'grid': e.get('Grid_Size')}
        out['workloads'][label] = w
    json.dump(out, open(os.path.join(P, 'r03_traffic.json'), 'w'), indent=1)
    print('   r03_traffic.json')
# K2 instruction mix
if os.path.isdir(os.path.join(O, 'k2_sq1')):
    subprocess.run([py, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(O, 'k2_pmc_summary.json'),
                    'k2_128_4096', os.path.join(O, 'k2_sq1'), os.path.join(O, 'k2_sq2')], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run([py, os.path.join(ROOT, 'tools', 'k2_pmc_profile.py'), os.path.join(O, 'k2_pmc_summary.json'),
                    'k2_128_4096', '4096', os.path.join(P, 'r03_k2_pmc.json')], check=True)
