"""Composed parallel-transmit route (rfgr2beff -> blochsim -> backward to rf / gr) at 64^3 x 1024 for the coil
counts in argv: the program that goes after `rocprofv3 ... --` for kernel stats / PMC traffic of the
parallel-transmit K0 kernels."""
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd  # noqa: F401
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
g = torch.Generator(device='cpu').manual_seed(5)
for nC in [int(x) for x in sys.argv[1:]]:
    rf0 = (0.05 * torch.randn((1, 2, nT, nC), generator=g)).to(dev)
    b1 = torch.randn((1, n ** 3, 2, nC), generator=g).to(dev) * 0.3
    for it in range(4):
        rf, gr = rf0.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        Mo = sims.blochsim(sp['M0'], beff, **kw)
        Mo.sum().backward()
        del beff, Mo
    torch.cuda.synchronize()
