"""Step time (K0 + the K1 that follows it, XCD-contiguous order) against K0's block geometry / store flavour
(dev knob MRPHY_K0_VARIANT = order*1000 + rows/8*10 + nt), interleaved.   python tools/k0var_step_ab.py OUT.json"""
import json, os, statistics, sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
build_dev.use()
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
K0V = ['0', '2020', '2041', '2081', '2161', '2321', '1021', '21']
res = []
for label, n, nM, nT in (('cfg1 64^3x1024', 64, 64 ** 3, 1024), ('shard 262144x4096', 128, 262144, 4096)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT + 36 * nM
    t0 = {c: [] for c in K0V}; t1 = {c: [] for c in K0V}
    with torch.no_grad():
        blk = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
        for rep in range(10):
            for c in K0V:
                os.environ['MRPHY_K0_VARIANT'] = c
                e = [ev() for _ in range(3)]
                e[0].record()
                beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
                e[1].record()
                sims.blochsim(sp['M0'], blk, **kw)
                e[2].record(); torch.cuda.synchronize()
                if rep:
                    t0[c].append(e[0].elapsed_time(e[1])); t1[c].append(e[1].elapsed_time(e[2]))
    for c in K0V:
        r = dict(size=label, K0_variant=c, K0_ms=round(statistics.median(t0[c]), 4), K1_ms=round(statistics.median(t1[c]), 4),
                 K1_frac=round(alg / (statistics.median(t1[c]) * 1e-3) / 8e12, 3),
                 step_ms=round(statistics.median(t0[c]) + statistics.median(t1[c]), 4))
        print(json.dumps(r), flush=True); res.append(r)
    del blk, sp
os.environ['MRPHY_K0_VARIANT'] = '0'
json.dump({'runs': res}, open(sys.argv[1], 'w'), indent=1)
