"""Does the adjoint kernel's time depend on the relative placement of Beff / history / grad_Beff
(three streams of identical size 3*2^33 B at 128^3 x 1024)?  Spacer allocations shift them."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import sims, synth, beffective  # noqa: E402

dev = torch.device('cuda:0')
n, nT = 128, 1024
sp = synth.cube_spins(n, device=dev)
pu = synth.pulse(nT, device=dev)
for spacer in (0, 0, 4 << 20, (4 << 20) + 4096, 256 << 20, (1 << 30) + (37 << 12)):
    torch.cuda.empty_cache()
    keep = []
    beff = beffective.rfgr2beff(pu['rf'], pu['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], lazy=False)
    if spacer:
        keep.append(torch.empty(spacer, dtype=torch.uint8, device=dev))
    Mi_ = sp['M0'].clone().requires_grad_(True)
    b_ = beff.detach().requires_grad_(True)
    ts = []
    for it in range(3):
        Mo = sims.blochsim(Mi_, b_, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=pu['dt'])
        if spacer and it == 0:
            keep.append(torch.empty(spacer, dtype=torch.uint8, device=dev))
        g = torch.ones_like(Mo)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); Mo.backward(g); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        gp = b_.grad.data_ptr()
        Mi_.grad = None; b_.grad = None
    print(f'spacer {spacer:>11d}: K3 {min(ts):.3f} ms   beff@{beff.data_ptr():#x} gBeff@{gp:#x} '
          f'delta {(gp - beff.data_ptr()) / 2**33:.6f} x 2^33', flush=True)
    del beff, b_, Mo, g, keep
