"""Condense gpurun_out/r02p (tools/runs/r02_profiles.sh) into the tracked profiles/r02_* files:
bench JSON lines, rocprofv3 kernel_stats (our kernels + top rows), our kernels' dispatches from the
kernel traces, and HBM traffic per launch from the PMC passes with the ratio to the algorithmic bytes.

    python tools/collect_profiles.py [gpurun_out/r02p] [r02]
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r02p')
tag = sys.argv[2] if len(sys.argv) > 2 else 'r02'
P = os.path.join(ROOT, 'profiles')


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


def copy_json(name, out):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f):
        line = [ln for ln in open(f).read().splitlines() if ln.startswith('{')][-1]
        json.loads(line)
        open(os.path.join(P, out), 'w').write(line + '\n')
        return json.loads(line)


def stats(dirname, out, keep=14):
    f = one(f'{dirname}/**/*kernel_stats.csv')
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(P, out), 'w', newline='') as o:
            csv.writer(o, quoting=csv.QUOTE_ALL).writerows(rows[:keep + 1])


def trace(dirname, out):
    f = one(f'{dirname}/**/*kernel_trace.csv')
    if f:
        rows = list(csv.DictReader(open(f)))
        mine = [r for r in rows if 'k_' in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
        cols = [c for c in ('Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Workgroup_Size', 'Grid_Size',
                            'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size')
                if c in rows[0]]
        with open(os.path.join(P, out), 'w', newline='') as o:
            w = csv.writer(o)
            w.writerow(cols + ['Duration_ms'])
            for r in mine:
                w.writerow([r[c] for c in cols] + [f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.4f}"])


def pmc(fetch_dir, write_dir, alg, out):
    a = os.path.join(src, f'_alg_{out}')
    json.dump(alg, open(a, 'w'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_bytes.py'), os.path.join(src, fetch_dir),
                        os.path.join(src, write_dir), a], capture_output=True, text=True)
    os.remove(a)
    if r.returncode == 0 and r.stdout.strip().startswith('{'):
        open(os.path.join(P, out), 'w').write(r.stdout)
        return json.loads(r.stdout)
    print('pmc failed', out, r.stderr[-500:])


rows = 128 ** 3
b = copy_json('bench_fwd.json', f'{tag}_bench_n128_nT4096.json')
stats('prof_fwd', f'{tag}_bench_n128_nT4096_kernel_stats.csv')
trace('prof_fwd', f'{tag}_bench_n128_nT4096_kernel_trace_mrphy.csv')
nT = 4096
t = pmc('pmc_fetch_fwd', 'pmc_write_fwd',
        {'k_bloch_fwd_lines': 12 * rows * nT + rows * 36, 'k_rfgr2beff<': 12 * rows * nT + rows * 16},
        f'{tag}_bench_n128_nT4096_pmc_hbm.json')
if t:
    k1 = next(v for k, v in t.items() if 'k_bloch_fwd_lines' in k)
    k0 = next(v for k, v in t.items() if 'k_rfgr2beff<' in k)
    json.dump({'note': 'HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in '
               'separate runs; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 rule for 16 B/lane '
               f'coalesced reads), workload 128^3 x 4096 fp32, round {tag}',
               'k_bloch_fwd_bytes_per_launch': k1['total'], 'k_bloch_fwd_fetch_bytes': k1['fetch_bytes_x2'],
               'k_bloch_fwd_write_bytes': k1['write_bytes'], 'k_rfgr2beff_bytes_per_launch': k0['total'],
               'k_rfgr2beff_write_bytes': k0['write_bytes']}, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
copy_json('bench_grad128.json', f'{tag}_bench_grad_n128_nT1024.json')
stats('prof_grad128', f'{tag}_grad_n128_nT1024_kernel_stats.csv')
trace('prof_grad128', f'{tag}_grad_n128_nT1024_kernel_trace_mrphy.csv')
nT = 1024
pmc('pmc_fetch_grad128', 'pmc_write_grad128',
    {'k_bloch_bwd_lines': 36 * rows * nT + rows * 24, 'k_bloch_fwd_lines': 24 * rows * nT + rows * 36,
     'k_rfgr2beff<': 12 * rows * nT + rows * 16, 'k_rfgr2beff_bwd_p1v': 12 * rows * nT},
    f'{tag}_grad_n128_nT1024_pmc_hbm.json')
copy_json('bench_grad_cfg4.json', f'{tag}_bench_grad_cfg4_n64_nT2048.json')
stats('prof_grad_cfg4', f'{tag}_grad_cfg4_n64_nT2048_kernel_stats.csv')
print(open(os.path.join(src, 'kstats.txt')).read() if os.path.exists(os.path.join(src, 'kstats.txt')) else '')
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))
