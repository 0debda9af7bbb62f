"""cProfile of the host side of one fused forward call / one fwd+bwd iteration (32^3 x 512)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import fused, synth, beffective, sims  # noqa: E402

dev = torch.device('cuda:0')
sp = synth.cube_spins(32, device=dev)
p = synth.pulse(512, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])


def fwd():
    return fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)


def two():
    b = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    return sims.blochsim(sp['M0'], b, **kw)


def grad():
    rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
    Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
    Mo.sum().backward()
    return rf.grad


which = {'fwd': fwd, 'two': two, 'grad': grad}[sys.argv[1] if len(sys.argv) > 1 else 'fwd']
for _ in range(20):
    which()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    which()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
