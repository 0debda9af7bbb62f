"""Fewer resident waves = fewer concurrent row streams: do the line kernels like that?  K1 (fast, both
builds), K1h, K3 at CUBE^3 x NT with the workgroups per CU capped through dynamic LDS padding (dev
build, MRPHY_LDS_PAD): 16 / 12 / 10 / 8 waves per CU.
    python tools/occupancy_cap_sweep.py CUBE NT OUT.json [fwd]"""
import json
import os
import sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev  # noqa: E402
build_dev.use()
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402
n, nT = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ss = n ** 3 * nT
PADS = {16: 0, 12: 4352, 10: 7168, 8: 11264}          # 9216 + pad <= 160 KiB / waves


def t_of(fn, reps=4):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:]), out


res = []
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    sims.blochsim(sp['M0'], beff, **kw)
    for mode in ('precise', 'fast'):
        for fv in (331, 441):
            for cap, pad in PADS.items():
                os.environ.update(MRPHY_FWD_VARIANT=str(fv), MRPHY_LDS_PAD=str(pad))
                with mrphy_amd.precision(mode):
                    ms, _ = t_of(lambda: sims.blochsim(sp['M0'], beff, **kw))
                r = dict(kernel='K1', mode=mode, build=fv, waves_per_cu_cap=cap, ms=round(ms, 4), frac=round(12 * ss / ms / 8e9, 3))
                print(json.dumps(r), flush=True); res.append(r)
os.environ.pop('MRPHY_FWD_VARIANT', None)
if not (len(sys.argv) > 4 and sys.argv[4] == 'fwd'):
    beff.requires_grad_(True)
    Mi = sp['M0'].clone().requires_grad_(True)
    for rnd in range(2):
        for cap, pad in PADS.items():
            os.environ['MRPHY_LDS_PAD'] = str(pad)
            tf, Mo = t_of(lambda: sims.blochsim(Mi, beff, **kw))
            tb, g = t_of(lambda: torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo), retain_graph=True))
            del g, Mo
            r = dict(kernel='K1h+K3', round=rnd, waves_per_cu_cap=cap, K1h_ms=round(tf, 4), K1h_frac=round(24 * ss / tf / 8e9, 3),
                     K3_ms=round(tb, 4), K3_frac=round(36 * ss / tb / 8e9, 3))
            print(json.dumps(r), flush=True); res.append(r)
os.environ['MRPHY_LDS_PAD'] = '0'
json.dump({'cube': n, 'nT': nT, 'runs': res}, open(sys.argv[3], 'w'), indent=1)
