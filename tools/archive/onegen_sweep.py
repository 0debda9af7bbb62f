"""One-generation grids: K1 (both precision modes), K1h and K3 at CUBE^3 x NT under the dev
build's knobs -- occupancy variants, priority rotation -- with wave timelines.

    python tools/onegen_sweep.py CUBE NT [OUT.json] [fwd]      (fwd: the K1 part only)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tools')]
import build_dev  # noqa: E402
lib = build_dev.use()
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402

n, nT = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
tiles = (n ** 3 + 63) // 64
cap = (tiles + 7) // 8 * 8
stamps = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
ss = n ** 3 * nT


def spread():
    s = stamps.cpu().numpy().astype(np.int64)
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    st, en = (s[:, 0] - t0) * 1e-2, (s[:, 1] - t0) * 1e-2
    return {'end_us_min_med_max': [round(float(x), 0) for x in (en.min(), np.median(en), en.max())],
            'late_starts': int((st > 0.05 * en.max()).sum()),
            'mean_resident_frac': round(float((en - st).sum() / (en.max() * len(s))), 3)}


def timed(fn, reps=5):
    ts = []
    for i in range(reps):
        stamps.zero_()
        lib.mrphy_dev_set_stamps(stamps.data_ptr() if i == reps - 1 else None, cap)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    lib.mrphy_dev_set_stamps(None, 0)
    return out, min(ts[1:])


def setenv(**kv):
    for k_, v in kv.items():
        if v is None:
            os.environ.pop(k_, None)
        else:
            os.environ[k_] = str(v)


res = []
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    for mode in ('precise', 'fast'):
        for fv in ((None, 331, 441) if len(sys.argv) <= 5 else tuple(int(x) for x in sys.argv[5].split(","))):
            for pr in (0, 1):
                setenv(MRPHY_FWD_VARIANT=fv, MRPHY_PRIO_ROT=pr)
                with mrphy_amd.precision(mode):
                    _, ms = timed(lambda: sims.blochsim(sp['M0'], beff, **kw))
                r = dict(kernel='K1', mode=mode, fwd_variant=fv, prio_rot=pr, ms=round(ms, 4),
                         frac_hbm=round(12 * ss / ms / 8e9, 3), **spread())
                print(json.dumps(r), flush=True)
                res.append(r)
setenv(MRPHY_FWD_VARIANT=None, MRPHY_PRIO_ROT=0)
if len(sys.argv) > 4 and sys.argv[4] == 'fwd':
    json.dump({'cube': n, 'nT': nT, 'runs': res}, open(sys.argv[3], 'w'), indent=1)
    sys.exit(0)
beff.requires_grad_(True)
Mi = sp['M0'].clone().requires_grad_(True)
for mode in ('precise', 'fast'):
    for bv in (None, 2, 4):
        for pr in (0, 1):
            setenv(MRPHY_BWD_VARIANT=bv, MRPHY_PRIO_ROT=pr)
            with mrphy_amd.precision(mode):
                Mo, ms_f = timed(lambda: sims.blochsim(Mi, beff, **kw))
                sf = spread()
                g, ms_b = timed(lambda: torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo), retain_graph=True))
                sb = spread()
            del g, Mo
            r = dict(kernel='K1h+K3', mode=mode, bwd_variant=bv, prio_rot=pr, K1h_ms=round(ms_f, 4),
                     K1h_frac=round(24 * ss / ms_f / 8e9, 3), K3_ms=round(ms_b, 4),
                     K3_frac=round(36 * ss / ms_b / 8e9, 3), K1h=sf, K3=sb)
            print(json.dumps(r), flush=True)
            res.append(r)
if len(sys.argv) > 3:
    json.dump({'cube': n, 'nT': nT, 'runs': res}, open(sys.argv[3], 'w'), indent=1)
