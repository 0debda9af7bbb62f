"""History layout A/B on the SAME blocks in ONE process: hist[tile][t][xyz][lane] (shipped) against
hist[t][tile][xyz][lane] (-DMRPHY_HIST_TMAJOR dev build): K1h writing its history into each candidate block, K3 writing
grad_Beff into it (history elsewhere), both builds; results compared bit for bit first on a small problem.

    python tools/dbg/hist_layout_ab.py OUT.json A.so B.so
"""
import ctypes
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib, beffective, sims, synth  # noqa: E402
from mrphy_amd.workspace import _Pair  # noqa: E402

dev = torch.device('cuda', 0)


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.mrphy_abi_version() == _lib.ABI_VERSION
    return lib


libs = {'tile-major (shipped)': load(sys.argv[2]), 'time-major': load(sys.argv[3])}


def use(name):
    _lib._lib = libs[name]


ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def timed(fn, reps=2):
    fn()
    best = 1e9
    for _ in range(reps):
        a, b = ev(), ev()
        a.record(); fn(); b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(best, 4)


# 1. correctness on small problems (ragged last tile; line and chunked kernels; fp32 and fp64)
for n, nT, dt in ((12, 64, torch.float32), (11, 50, torch.float32), (12, 64, torch.float64), (9, 48, torch.float64)):
    sp = synth.cube_spins(n, dtype=dt, device=dev, seed_M0=3)
    p = synth.pulse(nT, dtype=dt, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    got = {}
    for name in libs:
        use(name)
        rf = p['rf'].clone().requires_grad_(True)
        M0 = sp['M0'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        beff.retain_grad()
        Mo = sims.blochsim(M0, beff, **kw)
        Mo.square().sum().backward()
        torch.cuda.synchronize()
        got[name] = (Mo.detach().clone(), M0.grad.clone(), beff.grad.clone(), rf.grad.clone())
    a, b = got.values()
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    print(f'small {n}^3 x {nT} {dt}: the two layouts give the same bits: {same}', flush=True)
    assert same

rows = []
for n, nT, nb in ((64, 2048, 6), (128, 1024, 4)):
    nM = n ** 3
    numel = nM * nT * 3
    field = torch.empty(numel, dtype=torch.float32, device=dev)
    beff = field.view(1, nM, nT, 3)
    beff.uniform_(-2.0, 2.0)
    beff.requires_grad_(True)
    other = torch.empty(numel, dtype=torch.float32, device=dev)
    Mi = torch.zeros((1, nM, 3), device=dev)
    Mi[..., 2] = 1
    gMo = torch.ones_like(Mi)
    blocks = [torch.empty(numel, dtype=torch.float32, device=dev) for _ in range(nb)]
    for i, b in enumerate(blocks):
        r = dict(cube=n, nT=nT, block=i)
        for name in libs:
            use(name)
            tH = timed(lambda: sims.blochsim(Mi, beff, workspace=_Pair(b, other)))
            Mo = sims.blochsim(Mi, beff, workspace=_Pair(other, b))
            tG = timed(lambda: torch.autograd.grad(Mo, beff, gMo, retain_graph=True))
            del Mo
            r[name] = dict(K1h_ms=tH, K3_ms=tG)
        print(json.dumps(r), flush=True)
        rows.append(r)
    del blocks, field, beff, other
    torch.cuda.empty_cache()
json.dump({'rows': rows}, open(sys.argv[1], 'w'), indent=1)
