"""Time the adjoint kernel with and without the grad_Beff output (same reads; no writes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd
from mrphy_amd import sims, synth, beffective

dev = torch.device('cuda:0')
n, nT = 128, 1024
sp = synth.cube_spins(n, device=dev)
pu = synth.pulse(nT, device=dev)
Mi = sp['M0']
beff = beffective.rfgr2beff(pu['rf'], pu['gr'], sp['loc'], lazy=False)
T1, T2 = sp['T1'], sp['T2']
for need_b in (True, False):
    Mi_ = Mi.clone().requires_grad_(True)
    b_ = beff.detach().requires_grad_(need_b)
    for it in range(3):
        Mo = sims.blochsim(Mi_, b_, T1=T1, T2=T2)
        g = torch.ones_like(Mo)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); Mo.backward(g); b.record(); torch.cuda.synchronize()
        Mi_.grad = None; b_.grad = None
    print('grad_Beff' if need_b else 'grad_Mi only', f'{a.elapsed_time(b):.3f} ms')
