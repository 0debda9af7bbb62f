"""Condense gpurun_out/<round>p (tools/runs/<round>_profiles.sh) into the tracked profiles/<round>_* files:
bench lines, rocprofv3 kernel-stats summaries -- PER KERNEL AND GRID SIZE (by_size below) --, PMC traffic (FETCH_SIZE /
WRITE_SIZE, separate passes, the guide's gfx950 corrections), the K2 instruction mix, the parity ledger.

    python tools/collect_profiles.py [gpurun_out/r06p [r06]]
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
O = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r06p')
R = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.normpath(O))[:3]       # 'r06'
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


def _short(name):
    return name.replace('void ', '').replace('(anonymous namespace)::', '').replace('mrphy::', '')


# `--stats` averages per kernel NAME, and one process launches K0 / K1 at several problem sizes
# (bench.py without --config: 128^3 x 4096, 64^3 x 1024, 64^3 x 2048; a workspace probe; warm-ups), so the r05 stats row
# of K1 averaged three sizes.  Here the kernel trace of the same run is split by (kernel, grid size): calls, mean,
# median, min -- and, for the launches whose workload is known, algorithmic bytes / mean / 8 TB/s.
#   kernel pattern -> {grid size in work-items: (label, algorithmic bytes per launch)}
def _k_bytes(nM, nT):
    return {'K0': 12 * nM * nT + 16 * nM, 'K1': 12 * nM * nT + 36 * nM,
            'K1h': 24 * nM * nT + 36 * nM, 'K3': 36 * nM * nT + 36 * nM}


def by_size(d, dst, workloads, note=''):
    r"""``workloads``: [(label, cube edge, nT, kernel families launched at that size)] of the profiled command.  Every (kernel, grid) group is reported; the
    roofline fraction only where the group's workload is unambiguous among `workloads` (the time-stepping kernels run one
    work-item per spin, K0 one per 256 B of ``Beff``)."""
    fs = glob.glob(os.path.join(O, d, '**', '*kernel_trace.csv'), recursive=True)
    if not fs:
        return
    groups = {}
    for r in csv.DictReader(open(fs[0])):
        n = _short(r['Kernel_Name'])
        if not n.startswith('k_'):
            continue
        g = int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])
        groups.setdefault((n.split('(')[0], g), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    fam = lambda n: ('K1h' if n.startswith('k_bloch_fwd_lines') and re.search(r', true, (true|false)>$', n) else  # noqa: E731
                     'K1' if n.startswith('k_bloch_fwd_lines') else 'K3' if n.startswith('k_bloch_bwd_lines') else
                     'K0' if n.startswith('k_rfgr2beff') and 'bwd' not in n else None)
    out = {'what': 'rocprofv3 --kernel-trace of the command below, launches grouped by (kernel, grid size in work-items); '
                   'frac_hbm = algorithmic bytes / mean duration / 8 TB/s where the workload of the group is known',
           'command': note, 'groups': []}
    for (n, g), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        ds.sort()
        e = {'kernel': n, 'grid_work_items': g, 'calls': len(ds), 'mean_ms': round(sum(ds) / len(ds), 4),
             'median_ms': round(ds[len(ds) // 2], 4), 'min_ms': round(ds[0], 4), 'max_ms': round(ds[-1], 4)}
        f = fam(n)
        # the time-stepping kernels run one work-item per spin (XCD-padded grids round the tile count up to 8)
        hits = [(lab, c, nT) for lab, c, nT, fams in workloads
                if f in ('K1', 'K1h', 'K3') and f in fams and (c ** 3 + 63) // 64 <= g // 64 <= (c ** 3 + 63) // 64 + 7]
        if '_f64' in n or n.startswith('k_rfgr2beff<double'):
            f = None                                  # the byte counts below are the fp32 ones
        if f == 'K0':       # K0: one work-item per 256 B of Beff (402 653 184 at 128^3 x 4096), i.e. grid = 12 nM nT / 256 (+ padding)
            hits = [(lab, c, nT) for lab, c, nT, fams in workloads if 'K0' in fams and abs(g - 12 * c ** 3 * nT // 256) <= 0.02 * g]
        if f and len(hits) == 1:
            lab, c, nT = hits[0]
            b = _k_bytes(c ** 3, nT)[f]
            e.update(workload=lab, family=f, algorithmic_bytes=b, frac_hbm_of_mean=round(b / (e['mean_ms'] * 1e-3) / 8e12, 4),
                     frac_hbm_of_median=round(b / (e['median_ms'] * 1e-3) / 8e12, 4))
        out['groups'].append(e)
    json.dump(out, open(os.path.join(P, dst), 'w'), indent=1)
    print('  ', dst)


# one collection per directory: the per-run CSVs carry the run id in their names, and a directory that
# gpurun merged two collections into would average (or, before pmc_summary keyed by file, SUM) them
for d in ('pmc_fetch_cfg2', 'pmc_write_cfg2', 'k2_sq1', 'k2b_sq1', 'pmc_fetch_cfg1', 'pmc_fetch_shard'):
    fs = glob.glob(os.path.join(O, d, '**', '*counter_collection.csv'), recursive=True)
    assert len(fs) <= 1, f'{d}: {len(fs)} counter files -- remove the local copy of {O} before a new collection'
for src, dst in (('bench_default.json', R + '_bench_default_all_configs.json'),
                 ('bench_cfg2.json', R + '_bench_cfg2_n128_nT4096_shard_of_8.json'),
                 ('bench_cfg1.json', R + '_bench_cfg1_n64_nT1024.json'),
                 ('bench_cfg4.json', R + '_bench_cfg4_grad_n64_nT2048.json'),
                 ('parity_ledger.json', R + '_parity.json'),
                 ('pytest_gpu_tail.txt', R + '_pytest_gpu_tail.txt'),
                 ('hist_policy.json', R + '_hist_parts_fresh_processes_final_tree.json')):
    cp(src, dst)
stats('prof_cfg2', R + '_bench_cfg2_kernel_stats.csv')
HEAD = ('128^3 x 4096 (configs[2], headline)', 128, 4096, ('K0', 'K1'))
CFG1 = ('64^3 x 1024 (configs[1])', 64, 1024, ('K0', 'K1'))
CFG4 = ('64^3 x 2048 (configs[4])', 64, 2048, ('K0', 'K1h', 'K3'))
by_size('prof_cfg2', R + '_bench_cfg2_by_size.json', [HEAD],
        'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-extra-configs')
by_size('prof_default', R + '_bench_default_by_size.json', [HEAD, CFG1, CFG4],
        'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu   (the driver\'s command: '
        'configs[1] and configs[4] timed in the same process; the K1h / K3 groups hold both routes of configs[4] and the '
        'workspace probe\'s launches are the <prec_f64, false, ...> instances)')
by_size('prof_cfg1', R + '_bench_cfg1_by_size.json', [CFG1],
        'rocprofv3 --kernel-trace --stats -- python3 bench.py --config 1 --steps 20 --warmup 2 --no-cpu')
by_size('prof_cfg4', R + '_bench_cfg4_by_size.json', [CFG4],
        'rocprofv3 --kernel-trace --stats -- python3 bench.py --config 4 --steps 30 --warmup 2 --no-cpu --grad-route allocator')
by_size('prof_cfg4_ws', R + '_bench_cfg4_workspace_by_size.json', [CFG4],
        'rocprofv3 --kernel-trace --stats -- python3 bench.py --config 4 --steps 30 --warmup 2 --no-cpu --grad-route workspace')
stats('prof_cfg1', R + '_bench_cfg1_kernel_stats.csv')
stats('prof_shard', R + '_shard_of_8_kernel_stats.csv')
stats('prof_cfg4', R + '_bench_cfg4_kernel_stats.csv')
stats('prof_f64', R + '_f64_grad_n64_nT1024_kernel_stats.csv')
stats('prof_f64fwd', R + '_f64_fwd_n64_nT1024_kernel_stats.csv')

# PMC traffic: pmc_summary.py merges the FETCH and WRITE passes per workload
py = sys.executable
for label, dirs in (('fwd_128_4096', ('pmc_fetch_cfg2', 'pmc_write_cfg2')),
                    ('fwd_64_1024', ('pmc_fetch_cfg1', 'pmc_write_cfg1')),
                    ('fwd_shard_262144_4096', ('pmc_fetch_shard', 'pmc_write_shard')),
                    ('grad_64_2048', ('pmc_fetch_cfg4', 'pmc_write_cfg4')),
                    ('fwd_f64_64_1024', ('pmc_fetch_f64fwd', 'pmc_write_f64fwd')),
                    ('grad_f64_64_1024', ('pmc_fetch_f64grad', 'pmc_write_f64grad'))):
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
    json.dump(out, open(os.path.join(P, R + '_traffic.json'), 'w'), indent=1)
    print('  ', R + '_traffic.json')
# K2 / K2b instruction mix (with the identifier of the kernel sources: bench.py uses them only when it matches)
if os.path.isdir(os.path.join(O, 'k2_sq1')):
    subprocess.run([py, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(O, 'k2_pmc_summary.json'),
                    'k2_128_4096', os.path.join(O, 'k2_sq1'), os.path.join(O, 'k2_sq2')], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run([py, os.path.join(ROOT, 'tools', 'k2_pmc_profile.py'), os.path.join(O, 'k2_pmc_summary.json'),
                    'k2_128_4096', '4096', os.path.join(P, R + '_k2_pmc.json')], check=True)
if os.path.isdir(os.path.join(O, 'k2b_sq1')):
    subprocess.run([py, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(O, 'k2b_pmc_summary.json'),
                    'gradfused_64_2048', os.path.join(O, 'k2b_sq1'), os.path.join(O, 'k2b_sq2')], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run([py, os.path.join(ROOT, 'tools', 'k2_pmc_profile.py'), os.path.join(O, 'k2b_pmc_summary.json'),
                    'gradfused_64_2048', '2048', os.path.join(P, R + '_k2b_pmc.json'), 'k_bloch_rfgr_bwd<', '4096'], check=True)
