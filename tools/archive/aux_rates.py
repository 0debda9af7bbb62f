"""Achieved HBM rate of the auxiliary kernels (the helpers either side of the hot path) at 128^3 spins,
fp32: algorithmic bytes / event time.  A coarse screen for kernels that sit far below what a streaming
kernel gets (6-7 TB/s):   python tools/aux_rates.py OUT.json"""
import json
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, sims, slowsims, utils, masks, synth  # noqa: E402
dev = torch.device('cuda', 0)
n = 128
nM = n ** 3
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
g = torch.Generator(device='cpu').manual_seed(1)
rnd = lambda *s: torch.rand(s, generator=g).to(dev)  # noqa: E731
res = []


def t_of(fn, reps=6):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts[1:]) // 2], out


def row(name, nbytes, fn):
    ms, out = t_of(fn)
    r = dict(op=name, ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1), bytes=nbytes)
    print(json.dumps(r), flush=True); res.append(r)
    return out


M, b = sp['M0'], rnd(1, nM, 3) * 2 - 1
T1, T2, df = sp['T1'], sp['T2'], sp['Δf']
dur = torch.tensor([1e-3], device=dev)
with torch.no_grad():
    row('sims.freeprec fwd', nM * (12 + 12 + 12), lambda: sims.freeprec(M, dur, T1=T1, T2=T2, Δf=df))
    g2 = torch.tensor(2 * 3.141592653589793 * 4257.6 * 4e-6, device=dev)
    U, Phi = row('beff2uϕ', nM * (12 + 12 + 4), lambda: beffective.beff2uϕ(b, g2))
    row('uϕrot', nM * (12 + 4 + 12 + 12), lambda: utils.uϕrot(U, Phi, M))
    E1, E2 = torch.exp(-4e-6 / T1), torch.exp(-4e-6 / T2)
    row('blochsim_1step', nM * (12 + 12 + 12 + 12), lambda: slowsims.blochsim_1step(M, M, b, E1, E1 - 1, E2, g2.expand(1, 1))[0])
    mask = torch.ones((1, n, n, n), dtype=torch.bool, device=dev)
    mask[0, ::3] = False
    ix = masks.MaskIndex(mask)
    v = rnd(1, n, n, n, 3)
    v_ = row('masks.extract (2/3 of 128^3 x 3)', ix.nM * (12 + 12 + 4), lambda: masks.extract(v, ix))
    row('masks.embed', ix.nM * (12 + 4) + nM * 12, lambda: masks.embed(v_, ix))
    fov, ofst = torch.tensor([[24., 24., 24.]], device=dev), torch.zeros((1, 3), device=dev)
    row('masks.cube_loc', ix.nM * (12 + 4), lambda: masks.cube_loc(ix, fov, ofst))
# gradients of the elementwise helpers
Mg = M.clone().requires_grad_(True)
out = sims.freeprec(Mg, dur, T1=T1, T2=T2, Δf=df)
row('sims.freeprec bwd', nM * (12 + 12 + 12), lambda: torch.autograd.grad(out, Mg, torch.ones_like(out), retain_graph=True))
bg = b.clone().requires_grad_(True)
U, Phi = beffective.beff2uϕ(bg, g2)
row('beff2uϕ bwd', nM * (12 + 12 + 4 + 12), lambda: torch.autograd.grad((U, Phi), bg, (torch.ones_like(U), torch.ones_like(Phi)), retain_graph=True))
Ug, Pg, Vg = (x.detach().clone().requires_grad_(True) for x in (U, Phi, M))
Vo = utils.uϕrot(Ug, Pg, Vg)
row('uϕrot bwd', nM * (12 + 4 + 12 + 12 + 12 + 4 + 12), lambda: torch.autograd.grad(Vo, (Ug, Pg, Vg), torch.ones_like(Vo), retain_graph=True))
# A/B propagation at 64^3 x 1024
n2, nT = 64, 1024
sp2 = synth.cube_spins(n2, dtype=torch.float32, device=dev, seed_M0=4)
p2 = synth.pulse(nT, dtype=torch.float32, device=dev)
with torch.no_grad():
    beff = beffective.rfgr2beff(p2['rf'], p2['gr'], sp2['loc'], Δf=sp2['Δf'], γ=sp2['γ'])
    E1, E2 = torch.exp(-p2['dt'] / sp2['T1']), torch.exp(-p2['dt'] / sp2['T2'])
    A, B = row('beff2ab 64^3 x 1024 (VALU: 4 columns)', n2 ** 3 * nT * 12, lambda: beffective.beff2ab(beff, E1=E1, E2=E2, γ=sp2['γ'], dt=p2['dt']))
    row('blochsim_ab', n2 ** 3 * (12 + 36 + 12 + 12), lambda: slowsims.blochsim_ab(sp2['M0'], A, B))
json.dump({'note': '128^3 spins fp32 unless noted; bytes = algorithmic', 'runs': res}, open(sys.argv[1], 'w'), indent=1)
