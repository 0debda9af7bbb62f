"""One-generation grids: the no-history K1 line kernel with 64-row tiles (4096 waves = 3072 + 1024 lone ones at
64^3) against 48-row tiles (5462 = 3072 + 2390), dev knob MRPHY_FWD_VARIANT = 364 / 348, interleaved; outputs
must be bit-identical (a spin's arithmetic does not depend on its tile).   python tools/onegen_rows_ab.py OUT.json"""
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
for n, nT, shard in ((64, 1024, 1), (64, 2048, 1), (64, 4096, 1), (128, 4096, 8), (80, 1024, 1), (96, 1024, 1), (128, 1024, 1)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    if shard > 1:                                   # rank 0's 1/8 block of the cube
        m = n ** 3 // shard
        sp = {k: (v[:, :m].contiguous() if torch.is_tensor(v) and v.ndim >= 2 and v.shape[1] == n ** 3 else v) for k, v in sp.items()}
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        for mode in ('precise', 'fast'):
            ts = {364: [], 348: []}
            outs = {}
            with mrphy_amd.precision(mode):
                for rep in range(11):
                    for v in (364, 348):
                        os.environ['MRPHY_FWD_VARIANT'] = str(v)
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize(); a.record()
                        outs[v] = sims.blochsim(sp['M0'], beff, **kw)
                        b.record(); torch.cuda.synchronize()
                        if rep:
                            ts[v].append(a.elapsed_time(b))
            spins = sp['M0'].shape[1]
            r = dict(cube=n, shard_of=shard, nT=nT, spins=spins, tiles64=(spins + 63) // 64, tiles48=(spins + 47) // 48, mode=mode,
                     rows64_ms=[round(statistics.median(ts[364]), 4), round(min(ts[364]), 4)],
                     rows48_ms=[round(statistics.median(ts[348]), 4), round(min(ts[348]), 4)],
                     bitwise_equal=bool(torch.equal(outs[364], outs[348])))
            r['rows48_over_rows64_median'] = round(r['rows48_ms'][0] / r['rows64_ms'][0], 3)
            r['frac_hbm_64_48'] = [round(12 * spins * nT / (r['rows64_ms'][0] * 1e-3) / 8e12, 3), round(12 * spins * nT / (r['rows48_ms'][0] * 1e-3) / 8e12, 3)]
            print(json.dumps(r), flush=True); res.append(r)
    del beff
os.environ.pop('MRPHY_FWD_VARIANT', None)
json.dump({'runs': res}, open(sys.argv[1], 'w'), indent=1)
