"""The program for the placement PMC passes: K1h and K3 at CUBE^3 x NT, TRIALS times with freshly
placed buffers (empty_cache + a spacer of varying size), so that one rocprofv3 pass sees both the
fast and the slow mode of each kernel.  python tools/placement_pmc_run.py CUBE NT TRIALS"""
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, sims, synth  # noqa: E402
n, nT, trials = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
for tr in range(trials):
    torch.cuda.empty_cache()
    spacer = torch.empty((tr * 1536 + 1) << 20, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    beff.requires_grad_(True)
    Mi = sp['M0'].clone().requires_grad_(True)
    for rep in range(2):
        Mo = sims.blochsim(Mi, beff, **kw)
        g = torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo))
        del g, Mo
    torch.cuda.synchronize()
    del beff, Mi, spacer
print('done')
