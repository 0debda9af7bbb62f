"""K0 (single coil) block-shape sweep at small sizes (dev build, MRPHY_K0_VARIANT = order*1000 + rows/8*10 + nt)."""
import json, os, statistics, sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
build_dev.use()
import mrphy_amd
from mrphy_amd import beffective, synth
dev = torch.device('cuda', 0)
for n, nT in ((64, 1024), (64, 2048), (128, 1024)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    res = {}
    with torch.no_grad():
        for rep in range(3):
            for v in (0, 2011, 2021, 2020, 2041, 2081, 2161, 11, 21, 41, 161):
                os.environ['MRPHY_K0_VARIANT'] = str(v)
                ts = []
                for _ in range(6):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); a.record()
                    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
                    b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                res.setdefault(v, []).append(statistics.median(ts[1:]))
    print(n, nT, {v: round(min(t), 4) for v, t in res.items()}, flush=True)
    del beff
