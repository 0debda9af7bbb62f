"""Does it matter WHICH eighth of a buffer each XCD sweeps?  For several fresh placements of the
buffers: K0, K1h, K3 at 128^3 x 1024 with the XCD slot -> region assignment rotated by 0..7 (dev build).
    python tools/xcd_shift_sweep.py OUT.json [trials]"""
import ctypes
import json
import sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev  # noqa: E402
lib = build_dev.use()
lib.mrphy_dev_set_xcd_shift.restype = ctypes.c_int
lib.mrphy_dev_set_xcd_shift.argtypes = [ctypes.c_int]
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, sims, synth  # noqa: E402
dev = torch.device('cuda', 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, nT = 128, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def t_of(fn):
    a, b = ev(), ev()
    torch.cuda.synchronize()
    a.record(); out = fn(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b), out


res = []
for tr in range(trials):
    torch.cuda.empty_cache()
    spacer = torch.empty((tr * 1536 + 1) << 20, dtype=torch.uint8, device=dev)
    row = {'trial': tr, 'K0': [], 'K1h': [], 'K3': []}
    beff = None
    for sh in list(range(8)) + [0]:
        assert lib.mrphy_dev_set_xcd_shift(sh) == 0
        with torch.no_grad():
            del beff
            ts = []
            for _ in range(3):
                t, b_ = t_of(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']))
                ts.append(t); beff = b_
            row['K0'].append(round(min(ts[1:]), 3))
        beff.requires_grad_(True)
        Mi = sp['M0'].clone().requires_grad_(True)
        tf, tb = [], []
        for _ in range(3):
            t, Mo = t_of(lambda: sims.blochsim(Mi, beff, **kw)); tf.append(t)
            t, g = t_of(lambda: torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo))); tb.append(t)
            del g, Mo
        row['K1h'].append(round(min(tf[1:]), 3)); row['K3'].append(round(min(tb[1:]), 3))
        beff = beff.detach()
    print(json.dumps(row), flush=True)
    res.append(row)
    del beff, spacer
lib.mrphy_dev_set_xcd_shift(0)
json.dump({'note': 'ms at 128^3 x 1024 for XCD shift 0..7 and 0 again (last entry), per fresh placement', 'runs': res},
          open(sys.argv[1], 'w'), indent=1)
