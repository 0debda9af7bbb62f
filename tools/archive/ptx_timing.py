"""Parallel transmit, forward + backward to rf / gr at 64^3 x 1024 per coil count: the route
fused.blochsim_rfgr takes (fused K2/K2b up to 8 coils, composed K0 + K1h + K3 + K0 adjoint beyond)
and the per-kernel split of the composed route.   python tools/ptx_timing.py OUT.json"""
import json
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402
dev = torch.device('cuda', 0)
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
g = torch.Generator(device='cpu').manual_seed(5)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
res = []
for nC in (1, 2, 8, 9, 16, 32):
    rf0 = (0.05 * torch.randn((1, 2, nT, nC), generator=g)).to(dev)
    b1 = torch.randn((1, n ** 3, 2, nC), generator=g).to(dev) * 0.3
    tot, parts = [], []
    for it in range(6):
        rf, gr = rf0.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        e = [ev() for _ in range(2)]
        e[0].record()
        Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ_beff=sp['γ'], **kw)
        Mo.sum().backward()
        e[1].record(); torch.cuda.synchronize()
        tot.append(e[0].elapsed_time(e[1]))
        # the composed route, stage by stage
        rf, gr = rf0.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        s = [ev() for _ in range(4)]
        s[0].record()
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        s[1].record()
        Mo = sims.blochsim(sp['M0'], beff, **kw)
        s[2].record()
        Mo.sum().backward()
        s[3].record(); torch.cuda.synchronize()
        parts.append([s[i].elapsed_time(s[i + 1]) for i in range(3)])
        del beff, Mo
    pm = [round(min(x[i] for x in parts[1:]), 3) for i in range(3)]
    r = dict(nC=nC, route_ms=round(min(tot[1:]), 3), composed_ms=round(sum(pm), 3), K0_ms=pm[0], K1h_ms=pm[1],
             backward_K3_K0adj_ms=pm[2])
    print(json.dumps(r), flush=True); res.append(r)
json.dump({'workload': '64^3 x 1024 fp32, forward + backward to rf / gr, b1 map', 'runs': res}, open(sys.argv[1], 'w'), indent=1)
