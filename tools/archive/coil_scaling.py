"""Coil-count scaling of the parallel-transmit kernels at 64^3 x 1024, fp32: K0 (rfgr2beff), K2 (fused forward),
and the composed forward + backward to rf / gr -- is there a cliff beyond 32 coils?
    python tools/coil_scaling.py OUT.json"""
import json, statistics, sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, fused, synth
dev = torch.device('cuda', 0)
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ev = lambda: torch.cuda.Event(enable_timing=True)
def t_of(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        a, b = ev(), ev(); torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return round(statistics.median(ts[1:]), 4)
res = []
g = torch.Generator(device='cpu').manual_seed(3)
for nC in (8, 16, 24, 32, 33, 40, 48, 64, 65, 96, 128):
    rf = (torch.rand((1, 2, nT, nC), generator=g) * 0.02).to(dev)
    b1 = torch.rand((1, n ** 3, 2, nC), generator=g).to(dev)
    with torch.no_grad():
        k0 = t_of(lambda: beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ']))
        k2 = t_of(lambda: fused.blochsim_rfgr(sp['M0'], rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ_beff=sp['γ'], **kw), 3 if nC > 64 else 5)
    def fb():
        r = rf.clone().requires_grad_(True)
        be = beffective.rfgr2beff(r, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        sims.blochsim(sp['M0'], be, **kw).sum().backward()
    fbt = t_of(fb, 3)
    r = dict(nC=nC, K0_ms=k0, K2_ms=k2, composed_fwd_bwd_ms=fbt)
    print(json.dumps(r), flush=True); res.append(r)
    del rf, b1
json.dump({'cube': n, 'nT': nT, 'runs': res}, open(sys.argv[1], 'w'), indent=1)
