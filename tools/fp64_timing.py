"""fp64 path (chunked kernels): rfgr2beff, blochsim forward, forward with history + adjoint."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float64, device=dev)
p = synth.pulse(nT, dtype=torch.float64, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ss = n ** 3 * nT


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


with torch.no_grad():
    t0 = timeit(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']))
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    t1 = timeit(lambda: sims.blochsim(sp['M0'], beff, **kw))
    t2 = timeit(lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw))
print(f'fp64 {n}^3 x {nT}: K0 {t0:.3f} ms ({24 * ss / t0 / 1e6:.0f} GB/s)  K1 {t1:.3f} ms '
      f'({24 * ss / t1 / 1e6:.0f} GB/s)  K2 fused {t2:.3f} ms ({ss / t2 / 1e6:.0f} G ss/s)')


def grad():
    b = beff.detach().requires_grad_(True)
    sims.blochsim(sp['M0'], b, **kw).sum().backward()


t3 = timeit(grad, 3)
print(f'fp64 fwd(history)+bwd {t3:.3f} ms ({(48 + 72) * ss / t3 / 1e6:.0f} GB/s algorithmic)')
