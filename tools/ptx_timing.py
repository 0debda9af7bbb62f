"""Parallel transmit, 64^3 x 1024: cost as a function of the number of coils, across the nC = 8
boundary of the register/LDS coil paths (K0_MAXC, K2_MAXC, K2B_MAXC in csrc/).  Per coil count:
K0 (rfgr2beff) and K2 (fused forward) alone, and one pulse-design iteration (forward + backward to
rf, gr) through the fused kernels and through the materialised path.

    python tools/ptx_timing.py [nC ...]        (default: 1 2 8 9 16)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')
n, nT = 64, 1024
coils = [int(a) for a in sys.argv[1:]] or [1, 2, 8, 9, 16]
sp = synth.cube_spins(n, device=dev)
p = synth.pulse(nT, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def t_avg(f, reps=6):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rel = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
print(f'{n}^3 x {nT}, precision {mrphy_amd.precision.get()}; ms')
print(f'{"nC":>3} {"K0":>8} {"K2 fwd":>8} {"fused f+b":>10} {"mater. f+b":>11}   grad_rf / grad_gr fused vs materialised')
for nC in coils:
    g = torch.Generator().manual_seed(1)
    rf0 = (p['rf'][..., None] * (0.5 + torch.rand(1, 1, 1, nC, generator=g)).to(dev)).contiguous()
    b1 = ((torch.rand(1, n ** 3, 2, nC, generator=g) * 2 - 1) * 0.4).to(dev)

    def step(kind):
        rf, gr = rf0.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        if kind == 'fused':
            Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ_beff=sp['γ'], **kw)
        else:
            be = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
            Mo = sims.blochsim(sp['M0'], be, **kw)
        Mo.sum().backward()
        return rf.grad, gr.grad

    with torch.no_grad():
        k0 = t_avg(lambda: beffective.rfgr2beff(rf0, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ']))
        k2 = t_avg(lambda: fused.blochsim_rfgr(sp['M0'], rf0, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1,
                                               γ_beff=sp['γ'], **kw))
    tf, tm = t_avg(lambda: step('fused')), t_avg(lambda: step('materialised'))
    gf, gm = step('fused'), step('materialised')
    print(f'{nC:3d} {k0:8.3f} {k2:8.3f} {tf:10.3f} {tm:11.3f}   {rel(gf[0], gm[0]):.1e} / {rel(gf[1], gm[1]):.1e}',
          flush=True)
