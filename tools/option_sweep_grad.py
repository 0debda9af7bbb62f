"""Forward+backward (to rf, gr) time of the materialised and the fused route over option
combinations at 64^3 x 1024 (and a batched shape), to spot cliffs.  ms per iteration."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print(f"{'N':>2} {'nM':>7} {'nT':>5} relax maps  b1  df pulseN | materialised   fused  (ms)")
for N, n, nT in ((1, 64, 1024), (4, 40, 1024)):
    sp = synth.cube_spins(n, device=dev)
    p = synth.pulse(nT, device=dev)
    nM = n ** 3
    ex = lambda x: x.expand((N,) + tuple(x.shape[1:])).contiguous()  # noqa: E731
    loc, M0 = ex(sp['loc']), ex(sp['M0'])
    for relax, maps, b1, df, pn in itertools.product((True, False), (True, False), (False, True),
                                                     (True, False), (1, N)):
        if (not relax and maps) or (pn != 1 and N == 1):
            continue
        T1 = (ex(sp['T1']) if maps else torch.tensor([[1.0]], device=dev)) if relax else None
        T2 = (ex(sp['T2']) if maps else torch.tensor([[0.05]], device=dev)) if relax else None
        b1m = torch.rand(N, nM, 2, device=dev) if b1 else None
        dfm = ex(sp['Δf']) if df else None
        kw = dict(T1=T1, T2=T2, γ=sp['γ'], dt=p['dt'])
        rf0 = p['rf'].expand(pn, 2, nT).contiguous()
        gr0 = p['gr'].expand(pn, 3, nT).contiguous()

        def mat():
            rf, gr = rf0.clone().requires_grad_(True), gr0.clone().requires_grad_(True)
            b = beffective.rfgr2beff(rf, gr, loc, Δf=dfm, b1Map=b1m, γ=sp['γ'])
            sims.blochsim(M0, b, **kw).sum().backward()

        def fus():
            rf, gr = rf0.clone().requires_grad_(True), gr0.clone().requires_grad_(True)
            fused.blochsim_rfgr(M0, rf, gr, loc, Δf=dfm, b1Map=b1m, γ_beff=sp['γ'], **kw).sum().backward()
        print(f'{N:2d} {nM:7d} {nT:5d} {relax!s:>5} {maps!s:>5} {b1!s:>5} {df!s:>5} {pn:6d} | '
              f'{timeit(mat):12.3f} {timeit(fus):7.3f}', flush=True)
