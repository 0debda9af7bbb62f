"""fp32 data with the reference's fp64 default γ / dt (dtype code F32_C64) vs fp32 constants."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd
from mrphy_amd import beffective, sims, fused, synth
dev = torch.device('cuda:0')
n, nT = 128, 1024
sp, p = synth.cube_spins(n, device=dev), synth.pulse(nT, device=dev)
def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    for name, kw in (('fp32 constants', dict(γ=sp['γ'], dt=p['dt'])), ('fp64 defaults ', {})):
        k1 = t(lambda: sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], **kw))
        k2 = t(lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'],
                                           T1=sp['T1'], T2=sp['T2'], **kw))
        print(f'{name}: K1 {k1:.3f} ms   K2 {k2:.3f} ms')
