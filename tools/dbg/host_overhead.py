"""Host-side cost of one call of the Python layer (no sync inside the loop: per-call wall time = max(host, GPU)),
and a cProfile of 300 calls on a problem small enough to be host-bound."""
import cProfile, pstats, sys, time, io
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, fused, synth
dev = torch.device('cuda', 0)
for n, nT in ((16, 256), (32, 1024), (64, 1024)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    calls = {
        'fused': lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw),
        'rfgr2beff': lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']),
    }
    beff = calls['rfgr2beff']()
    calls['blochsim'] = lambda: sims.blochsim(sp['M0'], beff, **kw)
    with torch.no_grad():
        for name, f in calls.items():
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(300):
                f()
            t_host = (time.perf_counter() - t) / 300
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t) / 300
            print(f'{n}^3 x {nT} {name:10s}: issue {t_host * 1e6:7.1f} us per call, with the GPU drained {t_all * 1e6:7.1f} us', flush=True)
sp = synth.cube_spins(16, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(256, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(300):
        fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22)
print(s.getvalue()[:3800])
