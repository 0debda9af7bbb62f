"""Parallel-transmit pulse-design step (8 coils), 64^3 x 1024: fused kernels vs materialised path."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

dev = torch.device('cuda:0')
n, nT, nC = 64, 1024, 8
sp = synth.cube_spins(n, device=dev)
p = synth.pulse(nT, device=dev)
g = torch.Generator().manual_seed(1)
rf0 = (p['rf'][..., None] * (0.5 + torch.rand(1, 1, 1, nC, generator=g)).to(dev)).contiguous()
b1 = ((torch.rand(1, n ** 3, 2, nC, generator=g) * 2 - 1) * 0.4).to(dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])


def step(kind):
    rf, gr = rf0.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
    if kind == 'fused':
        Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ_beff=sp['γ'], **kw)
    else:
        be = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        Mo = sims.blochsim(sp['M0'], be, **kw)
    Mo.sum().backward()
    return rf.grad, gr.grad


res = {}
for kind in ('fused', 'materialised'):
    for _ in range(3):
        res[kind] = step(kind)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        step(kind)
    b.record(); torch.cuda.synchronize()
    print(f'{kind:13s} {a.elapsed_time(b) / 10:.3f} ms per fwd+bwd iteration')
rel = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
print('grad_rf rel-L2 fused vs materialised', rel(res['fused'][0], res['materialised'][0]),
      ' grad_gr', rel(res['fused'][1], res['materialised'][1]))
