import sys, statistics, torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, synth
dev = torch.device('cuda', 0)
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
g = torch.Generator(device='cpu').manual_seed(3)
for nC in (1, 8, 64, 65, 66, 72, 73, 80, 96):
    rf = (torch.rand((1, 2, nT, nC), generator=g) * 0.02).to(dev)
    b1 = torch.rand((1, n ** 3, 2, nC), generator=g).to(dev)
    ts = []
    with torch.no_grad():
        for i in range(6):
            a, b = ev(), ev(); torch.cuda.synchronize(); a.record()
            beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(nC, round(statistics.median(ts[1:]), 3), flush=True)
