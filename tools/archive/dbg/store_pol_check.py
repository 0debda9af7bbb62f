"""Which K0 store policy leaves the K1 that follows slow?  (dev build; cfg1 and shard; K1 plain order)
last digit of the K0 variant: 0 plain | 1 nt | 2 sc1 nt | 3 sc1 | 4 sc0 sc1 | 5 sc0 sc1 nt | 6 sc0 nt | 7 sc0"""
import json, os, statistics, sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
build_dev.use()
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
os.environ['MRPHY_K1_XCD'] = '0'
for label, n, nM, nT in (('cfg1', 64, 64 ** 3, 1024), ('shard', 128, 262144, 4096)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    blk = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
    with torch.no_grad():
        for v in (('2020', '2021', '2022', '2025', '2026', '2021') if os.environ.get('SHORT') else ('2020', '2021', '2022', '2023', '2024', '2025', '2026', '2027', '2021', '2022')):
            os.environ['MRPHY_K0_VARIANT'] = v
            t0, t1, t2 = [], [], []
            for rep in range(8):
                e = [ev() for _ in range(4)]
                e[0].record(); beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
                e[1].record(); sims.blochsim(sp['M0'], blk, **kw)
                e[2].record(); sims.blochsim(sp['M0'], blk, **kw)
                e[3].record(); torch.cuda.synchronize()
                if rep >= 2:
                    t0.append(e[0].elapsed_time(e[1])); t1.append(e[1].elapsed_time(e[2])); t2.append(e[2].elapsed_time(e[3]))
            print(label, 'K0 variant', v, 'K0 %.4f  K1 first %.4f  second %.4f' % tuple(statistics.median(t) for t in (t0, t1, t2)), flush=True)
