import os, sys, torch
sys.path.insert(0, '/root/repo')
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda:0')
for n, nT in ((64, 1024), (128, 1024)):
    sp, p = synth.cube_spins(n, device=dev), synth.pulse(nT, device=dev)
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        for relax in (True, False):
            kw = dict(T1=sp['T1'], T2=sp['T2']) if relax else {}
            f = lambda: sims.blochsim(sp['M0'], beff, γ=sp['γ'], dt=p['dt'], **kw)
            f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize()
            print(f'{n}^3 x {nT} relax={relax} variant={os.environ.get("MRPHY_FWD_VARIANT","default")}: {a.elapsed_time(b)/10:.3f} ms')
        del beff
