"""K1 (no history) builds A/B on one box, interleaved: the shipped schedule (MRPHY_FWD_VARIANT 331: rot_apply
chains sunk below the next batches' guards, 146 VGPRs, 3 waves/SIMD) against the pinned builds (1321 / 1331 /
1341 = 5-6 / 3-4 / 2-3 step batches with pin_state after each batch: 102 / 90 / 80 VGPRs, LDS-bound at 17
waves per CU), each also capped through dynamic LDS padding (dev knob MRPHY_LDS_PAD).  Sizes: BASELINE
configs[1], configs[4]'s forward, a 1/8 shard of configs[2], 128^3 x 1024, configs[2].
    python tools/k1_pin_ab.py OUT.json [reps]"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
# (variant, LDS pad): pad 0 = whatever the registers / 9 KB of LDS allow; 4096 -> 12 per CU; 1024 -> 16 per CU
CASES = [(331, 0), (1331, 0), (1331, 1024), (1331, 4096), (1321, 0), (1321, 1024), (1341, 0), (1341, 1024)]
res = []
SIZES = [('cfg1 64^3x1024', 64 ** 3, 1024), ('cfg4 64^3x2048', 64 ** 3, 2048), ('shard 262144x4096', 262144, 4096),
         ('128^3x1024', 128 ** 3, 1024), ('cfg2 128^3x4096', 128 ** 3, 4096)]
if os.environ.get('K1AB_SIZES'):
    SIZES = [SIZES[int(i)] for i in os.environ['K1AB_SIZES'].split(',')]
for label, nM, nT in SIZES:
    n = round(nM ** (1 / 3))
    if n ** 3 == nM:
        sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    else:                                     # first nM spins of the 128^3 cube (rank 0's shard)
        sp = synth.cube_spins(128, torch.arange(nM), dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT + nM * 36
    with torch.no_grad():
        os.environ['MRPHY_FWD_VARIANT'] = '0'; os.environ['MRPHY_LDS_PAD'] = '0'
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        for mode in ('precise', 'fast'):
            ts = {c: [] for c in CASES}
            norms = {}
            with mrphy_amd.precision(mode):
                for rep in range(reps + 1):
                    for c in CASES:
                        os.environ['MRPHY_FWD_VARIANT'] = str(c[0]); os.environ['MRPHY_LDS_PAD'] = str(c[1])
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize(); a.record()
                        Mo = sims.blochsim(sp['M0'], beff, **kw)
                        b.record(); torch.cuda.synchronize()
                        if rep:
                            ts[c].append(a.elapsed_time(b))
                        else:
                            norms[c] = Mo.clone()
            same = all(torch.equal(norms[CASES[0]], v) for v in norms.values())
            r = dict(size=label, spins=nM, nT=nT, mode=mode, bitwise_equal=bool(same),
                     ms={f'{c[0]}+pad{c[1]}': [round(statistics.median(ts[c]), 4), round(min(ts[c]), 4)] for c in CASES})
            r['frac_of_8TBps_median'] = {k: round(alg / (v[0] * 1e-3) / 8e12, 3) for k, v in r['ms'].items()}
            print(json.dumps(r), flush=True); res.append(r)
    del beff, sp
os.environ['MRPHY_FWD_VARIANT'] = '0'; os.environ['MRPHY_LDS_PAD'] = '0'
json.dump({'device': torch.cuda.get_device_name(0), 'runs': res}, open(sys.argv[1], 'w'), indent=1)
