"""Wall time per call vs kernel time at pulse-design sizes: how much of an optimisation-loop
iteration is host plumbing (torch ops for the constants, allocations, ctypes)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')
for n, nT in ((16, 256), (32, 512), (64, 1024)):
    sp = synth.cube_spins(n, device=dev)
    p = synth.pulse(nT, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])

    def two_kernel():
        b = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        return sims.blochsim(sp['M0'], b, **kw)

    def fused_fwd():
        return fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'],
                                   γ_beff=sp['γ'], **kw)

    def fused_grad():
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
        Mo.sum().backward()
        return rf.grad

    for name, fn in (('rfgr2beff+blochsim', two_kernel), ('fused fwd', fused_fwd),
                     ('fused fwd+bwd', fused_grad)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        K = 50
        t = time.perf_counter()
        for _ in range(K):
            fn()
        t_issue = (time.perf_counter() - t) / K          # host time to issue one call
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t) / K
        print(f'{n:3d}^3 x {nT:4d}  {name:20s} host issue {t_issue * 1e6:8.1f} us/call   '
              f'wall {t_all * 1e6:8.1f} us/call', flush=True)
