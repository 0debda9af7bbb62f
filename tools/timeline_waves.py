"""Wave timeline of the line kernels (K1, K1h, K3) from the dev build's per-workgroup stamps:
when each of the grid's waves started and ended, on which XCD / CU / SIMD -- how many generations
the grid really took, how ragged its tail is, how unevenly the dispatcher loaded the CUs.

    python tools/timeline_waves.py CUBE NT [OUT.json]
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


def analyse(name, ms):
    s = stamps.cpu().numpy().astype(np.int64)
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    st, en = (s[:, 0] - t0) * 1e-2, (s[:, 1] - t0) * 1e-2       # us (100 MHz clock)
    hw = s[:, 2]
    xcc = (hw >> 32) & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simdid = cuid * 4 + simd
    dur = en - st
    total = en.max()
    ncu, nsimd = len(np.unique(cuid)), len(np.unique(simdid))
    per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(simdid, return_inverse=True)[1])
    late = st > 0.05 * total                                      # started after 5 % of the kernel: 2nd generation
    q = lambda x, ps=(0, 5, 25, 50, 75, 95, 100): [round(float(np.percentile(x, p_)), 1) for p_ in ps]  # noqa: E731
    # machine utilisation over time: waves resident at 20 sample points
    grid_t = np.linspace(0, total, 21)[1:-1]
    resident = [int(((st <= g) & (en > g)).sum()) for g in grid_t]
    r = {'kernel': name, 'event_ms': ms, 'waves': int(len(s)), 'span_us': round(float(total), 1),
         'cus_used': ncu, 'simds_used': nsimd,
         'waves_per_cu_minmax': [int(per_cu.min()), int(per_cu.max())],
         'waves_per_simd_hist': np.bincount(per_simd).tolist(),
         'late_start_waves': int(late.sum()),
         'start_us_pct': q(st), 'end_us_pct': q(en), 'dur_us_pct': q(dur),
         'end_by_xcc_mean_us': [round(float(en[xcc == x].mean()), 1) for x in range(8) if (xcc == x).any()],
         'end_by_xcc_max_us': [round(float(en[xcc == x].max()), 1) for x in range(8) if (xcc == x).any()],
         'resident_waves_at_5pct_steps': resident,
         'mean_resident_frac_of_peak': round(float(dur.sum() / (total * max(resident))), 3)}
    print(json.dumps(r), flush=True)
    return r


def timed(fn):
    stamps.zero_()
    lib.mrphy_dev_set_stamps(stamps.data_ptr(), cap)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    out = fn()
    b.record()
    torch.cuda.synchronize()
    lib.mrphy_dev_set_stamps(None, 0)
    return out, a.elapsed_time(b)


res = []
if os.environ.get('TIMELINE_WHAT') == 'k2':          # the fused forward kernel instead (round 4)
    from mrphy_amd import fused
    f = lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)  # noqa: E731
    with torch.no_grad():
        for mode, rot in (('precise', '0'), ('precise', '1'), ('fast', '0'), ('fast', '1')):
            os.environ['MRPHY_K2_TEAM'] = rot
            with mrphy_amd.precision(mode):
                for _ in range(3):
                    f()
                ts = []
                for _ in range(5):
                    _, ms = timed(f)
                    ts.append(ms)
                r = analyse(f'K2 {mode} team={rot}', ms)
                r['event_ms_all'] = [round(t_, 4) for t_ in ts]
                s_ = stamps.cpu().numpy().astype(np.int64)
                s_ = s_[s_[:, 1] > 0]
                hw = s_[:, 2]
                simdid = ((((hw >> 32) & 0xf) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 64 + ((hw >> 8) & 0xf) * 4 + ((hw >> 4) & 3)
                # per SIMD: when its last wave ended, and the sum of its waves' lifetimes
                last = {}
                for sid, st_, en_ in zip(simdid, s_[:, 0], s_[:, 1]):
                    a_, b_ = last.get(int(sid), (0, 0))
                    last[int(sid)] = (max(a_, int(en_)), b_ + 1)
                t0 = s_[:, 0].min()
                ends = np.array([(v[0] - t0) * 1e-2 for v in last.values()])
                cnt = np.array([v[1] for v in last.values()])
                r['simd_last_end_us_by_wave_count'] = {int(c): [round(float(ends[cnt == c].mean()), 1), int((cnt == c).sum())] for c in np.unique(cnt)}
                print(json.dumps({'simd_last_end_us_by_wave_count': r['simd_last_end_us_by_wave_count']}), flush=True)
                res.append(r)
    if len(sys.argv) > 3:
        json.dump({'cube': n, 'nT': nT, 'kernels': res}, open(sys.argv[3], 'w'), indent=1)
    sys.exit(0)
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    for _ in range(2):
        sims.blochsim(sp['M0'], beff, **kw)
    for mode in ('precise', 'fast'):
        with mrphy_amd.precision(mode):
            sims.blochsim(sp['M0'], beff, **kw)
            _, ms = timed(lambda: sims.blochsim(sp['M0'], beff, **kw))
            res.append(analyse(f'K1 {mode}', ms))
beff.requires_grad_(True)
Mi = sp['M0'].clone().requires_grad_(True)
for it in range(3):
    Mo, ms_f = timed(lambda: sims.blochsim(Mi, beff, **kw))
    if it == 2:
        res.append(analyse('K1h', ms_f))
    g, ms_b = timed(lambda: torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo)))
    if it == 2:
        res.append(analyse('K3', ms_b))
    del g, Mo
if len(sys.argv) > 3:
    json.dump({'cube': n, 'nT': nT, 'kernels': res}, open(sys.argv[3], 'w'), indent=1)
