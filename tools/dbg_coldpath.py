"""Cost of the large-angle (phi > pi per step) branch: the same workload with 1x and 12x gradients."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import mrphy_amd
from mrphy_amd import beffective, sims, fused, synth
dev = torch.device('cuda:0')
n, nT = 64, 1024
sp, p = synth.cube_spins(n, device=dev), synth.pulse(nT, device=dev)
def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
for scale in (1.0, 12.0):
    gr = p['gr'] * scale
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        phi = (beff.norm(dim=-1) * (2 * torch.pi * sp['γ'] * p['dt'])[..., None])
        frac = float((phi > torch.pi).float().mean())
        k1 = t(lambda: sims.blochsim(sp['M0'], beff, **kw))
        k2 = t(lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], gr, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw))
        Mo = sims.blochsim(sp['M0'], beff, **kw)
    import bloch_c as C
    g = 2 * torch.pi * sp['γ'] * p['dt']
    E1, E2 = torch.exp(-p['dt'] / sp['T1']), torch.exp(-p['dt'] / sp['T2'])
    with mrphy_amd.constants_on('cpu'):
        Mo = sims.blochsim_consts(sp['M0'], beff, γ2πdt=g, E1=E1, E1_1=E1 - 1, E2=E2)
    want = C.blochsim(sp['M0'].cpu(), beff.cpu(), consts=C.constants_from(g.cpu(), E1.cpu(), E2.cpu(), (E1 - 1).cpu(), N=1, nM=n ** 3))
    err = float((Mo.double().cpu() - want).norm() / want.norm())
    print(f'gradient x{scale:4.1f}: {100 * frac:5.1f} % of spin-steps beyond pi; max phi {float(phi.max()):.1f} rad; '
          f'K1 {k1:.3f} ms  K2 {k2:.3f} ms; rel-L2 vs fp64 C arithmetic {err:.2e}')
