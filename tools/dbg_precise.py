"""Which of K1 / K2 / chunked-K1 deviates in 'precise' mode?  Compares each with exact arithmetic
(oracle/bloch_c.c) and with each other on a 4096-spin subset of the headline workload."""
import sys
import torch
sys.path[:0] = ['.', 'oracle']
import mrphy_amd
from mrphy_amd import beffective, sims, fused, synth
import bloch_c as C
dev = torch.device('cuda', 0)
n, nT = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
idx = synth.subset_indices(n, 4096, seed=5)
spc, pc = synth.cube_spins(n, idx, dtype=torch.float32), synth.pulse(nT, dtype=torch.float32)
sp = {k: v.to(dev) for k, v in spc.items()}
p = {k: v.to(dev) for k, v in pc.items()}
with mrphy_amd.constants_on('cpu'):
    g, E1, E2, E1_1 = sims.relax_constants(spc['T1'], spc['T2'], spc['γ'], pc['dt'], 4, dev)
consts = dict(γ2πdt=g, E1=E1, E1_1=E1_1, E2=E2)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    want = C.blochsim(spc['M0'], beff.cpu(), consts=C.constants_from(g, E1, E2, E1_1, N=1, nM=idx.numel()))
    for mode in ('fast', 'precise'):
        with mrphy_amd.precision(mode):
            k1 = sims.blochsim_consts(sp['M0'], beff, **consts)                      # line kernel
            pad = torch.empty(beff.numel() + 1, device=dev)[1:].view_as(beff)        # unaligned -> chunked
            pad.copy_(beff)
            k1c = sims.blochsim_consts(sp['M0'], pad, **consts)
            k2 = fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'],
                                     consts=consts)
            # without relaxation
            k1n = sims.blochsim_consts(sp['M0'], beff, γ2πdt=g)
            k2n = fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'],
                                      consts=dict(γ2πdt=g))
        print(f'{mode:8s} vs exact: lines {rel(k1, want):.3e} chunked {rel(k1c, want):.3e} fused {rel(k2, want):.3e} | '
              f'lines==chunked {bool((k1 == k1c).all())} lines==fused {bool((k1 == k2).all())} '
              f'(differ in {int((k1 != k2).sum())} of {k1.numel()}, max {float((k1 - k2).abs().max()):.2e}) | '
              f'no-relax lines==fused {bool((k1n == k2n).all())}', flush=True)
