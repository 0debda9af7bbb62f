"""Where are the worst spins?  Elementwise |Mo - exact| of the HIP kernels (precise and fast step) and of the reference's own
fp32 outputs (golden rows) on the 4096-spin subsets of BASELINE configs[1], [2], [4], against exact (fp64) arithmetic on
the same fp32 field and constants -- with each spin's total rotation angle and the coherence of its field,
|sum_t b_t| / sum_t |b_t| (1 = the field never changes direction: every step makes the same rounding errors).

    python tools/elementwise_scan.py OUT.json
"""
import json
import sys

import numpy as np
import torch

sys.path[:0] = ['.', 'oracle', 'tests']
import bloch_oracle as O  # noqa: E402
import cases  # noqa: E402
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims  # noqa: E402
from util import golden, t, to_dev  # noqa: E402

DEV = torch.device('cuda:0')
out = {}
for cfg, name, pulse in ((1, 'big_cfg1_f32', None), (4, 'big_cfg4_f32', 'interp'), (2, 'big_cfg2_f32', None)):
    G = golden(name)
    idx, sp, p = cases.big_subset(cfg, torch.float32, 4096)
    if pulse:
        I = golden('interp_f32')
        p = dict(rf=t(I['rf']), gr=t(I['gr']), dt=t(I['dt']))
    consts = {k: t(G[f'const.{k}']) for k in ('γ2πdt', 'E1', 'E1_1', 'E2')}
    spd, pd = to_dev(sp, DEV), to_dev(p, DEV)
    cd = {k: v.to(DEV) for k, v in consts.items()}
    beff = beffective.rfgr2beff(pd['rf'], pd['gr'], spd['loc'], Δf=spd['Δf'], γ=spd['γ'])
    got = {'HIP precise': sims.blochsim_consts(spd['M0'], beff, **cd).cpu()}
    with mrphy_amd.precision('fast'):
        got['HIP fast'] = sims.blochsim_consts(spd['M0'], beff, **cd).cpu()
    got['reference sims'], got['reference slowsims'] = t(G['Mo_sims']), t(G['Mo_slow'])
    bo = beff.cpu()
    torch.set_num_threads(16)
    exact = O.blochsim_f64_arith(sp['M0'], bo, consts=consts)
    g = float(consts['γ2πdt'].reshape(-1)[0])
    b = bo[0].double() * g
    phi = b.norm(dim=-1)
    tot = phi.sum(1)
    coh = b.sum(1).norm(dim=-1) / tot
    nT = bo.shape[2]
    r = {'nT': nT, 'phi_max': float(phi.max())}
    for k, v in got.items():
        d = (v.double() - exact).abs()[0]
        e = d.max(1).values
        top = torch.argsort(e, descending=True)[:5].tolist()
        r[k] = {'rel_l2': float((v.double() - exact).norm() / exact.norm()), 'max_abs': float(e.max()),
                'per_component': [float(x) for x in d.max(0).values], 'median_abs': float(e.median()),
                'rows_above_3e-5': int((e > 3e-5).sum()), 'rows_above_1e-5': int((e > 1e-5).sum()),
                'max_err_over_angle_bound': float((e / (tot * 2 ** -24)).max()),
                'worst': [dict(row=i, cube_index=int(idx[i]), err=float(e[i]), total_angle=float(tot[i]),
                               coherence=float(coh[i]), z=float(sp['loc'][0, i, 2])) for i in top]}
        print(cfg, k, json.dumps(r[k])[:600], flush=True)
    out[f'cfg{cfg}'] = r
json.dump(out, open(sys.argv[1], 'w'), indent=1)
