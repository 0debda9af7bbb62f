"""The bench's step (rfgr2beff -> blochsim on the block it has just written) for K1 builds x tile orders, interleaved
on one box (dev build): MRPHY_FWD_VARIANT in {331 shipped schedule, 1321 / 1331 pinned} x MRPHY_K1_XCD in {0 plain
tile order, 1 XCD-contiguous, 2 XCD-contiguous reversed}.  Medians of `reps` steps per case.
    python tools/k0k1_step_ab.py OUT.json [reps]       (K0K1_SIZES=0,1,2,3 selects sizes)"""
import json
import os
import statistics
import sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev  # noqa: E402
build_dev.use()
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
SIZES = [('cfg1 64^3x1024', 64, 64 ** 3, 1024), ('shard 262144x4096', 128, 262144, 4096),
         ('128^3x1024', 128, 128 ** 3, 1024), ('cfg2 128^3x4096', 128, 128 ** 3, 4096)]
if os.environ.get('K0K1_SIZES'):
    SIZES = [SIZES[int(i)] for i in os.environ['K0K1_SIZES'].split(',')]
CASES = [(v, x) for v in ('331', '1321', '1331') for x in ('0', '1', '2')]
res = []
for label, n, nM, nT in SIZES:
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT + 36 * nM
    for mode in ('precise', 'fast'):
        t0 = {c: [] for c in CASES}; t1 = {c: [] for c in CASES}
        ref = None; same = True
        with torch.no_grad(), mrphy_amd.precision(mode):
            for rep in range(reps + 1):
                for c in CASES:
                    os.environ['MRPHY_FWD_VARIANT'], os.environ['MRPHY_K1_XCD'] = c
                    e = [ev() for _ in range(3)]
                    e[0].record()
                    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
                    e[1].record()
                    Mo = sims.blochsim(sp['M0'], beff, **kw)
                    e[2].record(); torch.cuda.synchronize()
                    del beff
                    if rep:
                        t0[c].append(e[0].elapsed_time(e[1])); t1[c].append(e[1].elapsed_time(e[2]))
                    elif ref is None:
                        ref = Mo.clone()
                    else:
                        same = same and torch.equal(ref, Mo)
        r = dict(size=label, mode=mode, bitwise_equal=bool(same), cases={
            f'build {c[0]} xcd {c[1]}': dict(K0_ms=round(statistics.median(t0[c]), 4), K1_ms=round(statistics.median(t1[c]), 4),
                                             K1_frac=round(alg / statistics.median(t1[c]) / 1e9 / 8000 * 1e3 / 1e3, 3),
                                             step_ms=round(statistics.median(t0[c]) + statistics.median(t1[c]), 4)) for c in CASES})
        for k_, v in r['cases'].items():
            v['K1_frac'] = round(alg / (v['K1_ms'] * 1e-3) / 8e12, 3)
        print(json.dumps(r), flush=True); res.append(r)
    del sp
    torch.cuda.empty_cache()
os.environ['MRPHY_FWD_VARIANT'] = '0'; os.environ['MRPHY_K1_XCD'] = '0'
json.dump({'device': torch.cuda.get_device_name(0), 'runs': res}, open(sys.argv[1], 'w'), indent=1)
