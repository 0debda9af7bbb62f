"""One-generation grids (64^3 = 4096 tiles; the 3-wave K1 build holds 3072): K1 precise, uncapped
(3072 + 1024 waves) against a cap of 8 waves per CU (2048 + 2048) through dynamic LDS padding (dev build,
MRPHY_LDS_PAD), interleaved ABAB, median and min of 12 launches each.   python tools/onegen_cap_ab.py OUT.json"""
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
res = []
for n, nT in ((64, 1024), (64, 2048), (64, 4096), (80, 1024), (96, 1024)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        for mode in ('precise', 'fast'):
            ts = {0: [], 11264: []}
            with mrphy_amd.precision(mode):
                for rep in range(13):
                    for pad in (0, 11264):
                        os.environ['MRPHY_LDS_PAD'] = str(pad)
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize(); a.record()
                        sims.blochsim(sp['M0'], beff, **kw)
                        b.record(); torch.cuda.synchronize()
                        if rep:
                            ts[pad].append(a.elapsed_time(b))
            tiles = (n ** 3 + 63) // 64
            r = dict(cube=n, nT=nT, tiles=tiles, mode=mode,
                     uncapped_ms=[round(statistics.median(ts[0]), 4), round(min(ts[0]), 4)],
                     cap8_ms=[round(statistics.median(ts[11264]), 4), round(min(ts[11264]), 4)])
            r['cap8_over_uncapped_median'] = round(r['cap8_ms'][0] / r['uncapped_ms'][0], 3)
            print(json.dumps(r), flush=True); res.append(r)
    del beff
os.environ['MRPHY_LDS_PAD'] = '0'
json.dump({'runs': res}, open(sys.argv[1], 'w'), indent=1)
