"""Step time (K0 + the K1 that follows it) against K0's store policy (dev knob MRPHY_K0_VARIANT = order*1000 +
rows/8*10 + policy; policy 1 = nt, 2 = sc1 nt) and K1's tile order (MRPHY_K1_XCD), interleaved, seven sizes.   python tools/k0var_step_ab.py OUT.json"""
import json, os, statistics, sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
build_dev.use()
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
K0V = [(k, x) for k in ('2021', '2022') for x in ('1', '0')]        # K0 store policy nt | sc1 nt  x  K1 tile order XCD-contiguous | plain
res = []
for label, n, nM, nT in (('64^3x512', 64, 64 ** 3, 512), ('cfg1 64^3x1024', 64, 64 ** 3, 1024), ('cfg4 64^3x2048', 64, 64 ** 3, 2048),
                         ('shard 262144x4096', 128, 262144, 4096), ('128^3x1024', 128, 128 ** 3, 1024), ('128^3x2048', 128, 128 ** 3, 2048),
                         ('cfg2 128^3x4096', 128, 128 ** 3, 4096)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT + 36 * nM
    t0 = {c: [] for c in K0V}; t1 = {c: [] for c in K0V}
    with torch.no_grad():
        blk = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
        for rep in range(10):
            for c in K0V:
                os.environ['MRPHY_K0_VARIANT'], os.environ['MRPHY_K1_XCD'] = c
                e = [ev() for _ in range(3)]
                e[0].record()
                beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
                e[1].record()
                sims.blochsim(sp['M0'], blk, **kw)
                e[2].record(); torch.cuda.synchronize()
                if rep:
                    t0[c].append(e[0].elapsed_time(e[1])); t1[c].append(e[1].elapsed_time(e[2]))
    for c in K0V:
        r = dict(size=label, K0_variant=c[0], K1_xcd=c[1], K0_ms=round(statistics.median(t0[c]), 4), K1_ms=round(statistics.median(t1[c]), 4),
                 K1_frac=round(alg / (statistics.median(t1[c]) * 1e-3) / 8e12, 3),
                 step_ms=round(statistics.median(t0[c]) + statistics.median(t1[c]), 4))
        print(json.dumps(r), flush=True); res.append(r)
    del blk, sp
os.environ['MRPHY_K0_VARIANT'] = '0'; os.environ['MRPHY_K1_XCD'] = '0'
json.dump({'runs': res}, open(sys.argv[1], 'w'), indent=1)
