"""K0 alone: rows per block x store policy (dev build; MRPHY_K0_VARIANT = order*1000 + rows/8*10 + policy), cfg1 and shard."""
import os, statistics, sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
build_dev.use()
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for label, n, nM, nT in (('cfg1', 64, 64 ** 3, 1024), ('shard', 128, 262144, 4096)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    blk = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
    with torch.no_grad():
        for rows in (8, 16, 32, 64, 16):
            line = f'{label} rows {rows:3d}:'
            for pol in (1, 2, 1, 2):
                os.environ['MRPHY_K0_VARIANT'] = str(2000 + rows // 8 * 10 + pol)
                t0, t1 = [], []
                for rep in range(8):
                    e = [ev() for _ in range(3)]
                    e[0].record(); beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
                    e[1].record(); sims.blochsim(sp['M0'], blk, **kw)
                    e[2].record(); torch.cuda.synchronize()
                    if rep >= 2:
                        t0.append(e[0].elapsed_time(e[1])); t1.append(e[1].elapsed_time(e[2]))
                line += f'  pol {pol}: K0 {statistics.median(t0):.4f} K1 {statistics.median(t1):.4f}'
            print(line, flush=True)
