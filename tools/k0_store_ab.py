"""K0's store policy, judged by the STEP (K0 + K1) through the plain reference signatures, in a fresh process.

    python tools/k0_store_ab.py OUT.json          (appends one record per call: run it several times, fresh process each)

Per size: rfgr2beff(rf, gr, loc, Δf=, γ=[, store=policy]) -> sims.blochsim(Mi, Beff, ...), a fresh Beff per step from the
caching allocator (no out=), policies alternating (nt, sc1nt, nt, sc1nt, ...: 3 rounds of 6 steps each, first step of a
round dropped), K0 / K1 by HIP events, the step = their sum.
"""
import json
import os
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, sims, synth  # noqa: E402
from mrphy_amd.dist import shard_bounds  # noqa: E402

dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
SIZES = (('cfg1 64^3 x 1024', 64, 1024, None), ('1/8 shard of 128^3 x 4096', 128, 4096, 8), ('128^3 x 1024', 128, 1024, None))
rec = {'pid': os.getpid(), 'sizes': {}}
for name, n, nT, shard in SIZES:
    idx = None
    if shard:
        lo, hi = shard_bounds(n ** 3, shard, 0)
        idx = torch.arange(lo, hi, device=dev)
    sp = synth.cube_spins(n, idx, dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    acc = {'nt': [], 'sc1nt': []}
    with torch.no_grad():
        for rnd in range(3):
            for pol in ('nt', 'sc1nt'):
                for it in range(6):
                    e = [ev() for _ in range(3)]
                    e[0].record()
                    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], store=pol)
                    e[1].record()
                    Mo = sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
                    e[2].record()
                    torch.cuda.synchronize()
                    if it:
                        acc[pol].append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
                    del beff, Mo
    out = {}
    for pol, v in acc.items():
        k0 = sum(a for a, _ in v) / len(v)
        k1 = sum(b for _, b in v) / len(v)
        out[pol] = {'K0_ms': round(k0, 4), 'K1_ms': round(k1, 4), 'step_ms': round(k0 + k1, 4)}
    out['sc1nt_over_nt_step'] = round(out['sc1nt']['step_ms'] / out['nt']['step_ms'], 4)
    rec['sizes'][name] = out
    print(name, json.dumps(out), flush=True)
    del sp, p
    torch.cuda.empty_cache()
path = sys.argv[1]
allr = json.load(open(path)) if os.path.exists(path) else {'note': __doc__.strip().splitlines()[0], 'processes': []}
allr['processes'].append(rec)
json.dump(allr, open(path, 'w'), indent=1)
