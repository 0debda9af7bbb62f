"""Throughput of the three forward entry points over option combinations, to spot cliffs:
G spin-steps/s of rfgr2beff (K0), blochsim (K1), fused (K2) at ~2.7e8 spin-steps each."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


print(f"{'N':>2} {'nM':>7} {'nT':>5} relax maps  b1  df | {'K0':>7} {'K1':>7} {'K2':>7}  G spin-steps/s")
for N, n, nT in ((1, 64, 1024), (4, 40, 1024), (16, 25, 1024), (1, 32, 8192)):
    sp = synth.cube_spins(n, device=dev)
    p = synth.pulse(nT, device=dev)
    nM = n ** 3
    ex = lambda x: x.expand((N,) + tuple(x.shape[1:])).contiguous()  # noqa: E731
    loc, M0 = ex(sp['loc']), ex(sp['M0'])
    rf, gr = ex(p['rf']), ex(p['gr'])
    for relax, maps, b1, df in itertools.product((True, False), (True, False), (False, True), (True, False)):
        if not relax and maps:
            continue
        T1 = (ex(sp['T1']) if maps else torch.tensor([[1.0]], device=dev)) if relax else None
        T2 = (ex(sp['T2']) if maps else torch.tensor([[0.05]], device=dev)) if relax else None
        b1m = torch.rand(N, nM, 2, device=dev) if b1 else None
        dfm = ex(sp['Δf']) if df else None
        kw = dict(T1=T1, T2=T2, γ=sp['γ'], dt=p['dt'])
        with torch.no_grad():
            t0 = timeit(lambda: beffective.rfgr2beff(rf, gr, loc, Δf=dfm, b1Map=b1m, γ=sp['γ']))
            beff = beffective.rfgr2beff(rf, gr, loc, Δf=dfm, b1Map=b1m, γ=sp['γ'])
            t1 = timeit(lambda: sims.blochsim(M0, beff, **kw))
            t2 = timeit(lambda: fused.blochsim_rfgr(M0, rf, gr, loc, Δf=dfm, b1Map=b1m, γ_beff=sp['γ'], **kw))
            del beff
        ss = N * nM * nT
        print(f'{N:2d} {nM:7d} {nT:5d} {relax!s:>5} {maps!s:>5} {b1!s:>5} {df!s:>5} | '
              f'{ss / t0 / 1e9:7.1f} {ss / t1 / 1e9:7.1f} {ss / t2 / 1e9:7.1f}', flush=True)
