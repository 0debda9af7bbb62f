"""Whose error is it?  Gradients of L = sum(Mo) w.r.t. rf, gr, M0 on the seeded subsets of the
BASELINE configs: HIP two-kernel route, HIP fused route (both precision modes), the reference's
golden gradients (config 5) -- each against exact (fp64) differentiation of the same function
with the same fp32 constants (oracle/bloch_c.c).  Also splits the two-kernel route into its
stages (K3's grad_Beff vs exact; the K0 adjoint's spin sum vs an fp64 sum of the same grad_Beff).

    python tools/grad_parity.py [out.json]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import bloch_c as C  # noqa: E402
import cases  # noqa: E402
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402
from util import golden, t, rel_l2, to_dev  # noqa: E402

DEV = torch.device('cuda:0')
dev = lambda x: x.to(DEV)  # noqa: E731
out = {}


def gconsts(G, device=DEV):
    return {k: t(G[f'const.{k}']).to(device) for k in ('γ2πdt', 'E1', 'E1_1', 'E2')}


def hip_grads(sp, pulse, consts, route):
    spd = to_dev(sp, DEV)
    rf, gr = dev(pulse['rf']).requires_grad_(True), dev(pulse['gr']).requires_grad_(True)
    M0 = spd['M0'].clone().requires_grad_(True)
    extra = {}
    if route == 'two':
        beff = beffective.rfgr2beff(rf, gr, spd['loc'], Δf=spd['Δf'], γ=spd['γ'])
        beff.retain_grad()
        Mo = sims.blochsim_consts(M0, beff, **consts)
        Mo.sum().backward()
        extra = dict(beff=beff.detach(), gBeff=beff.grad)
    else:
        Mo = fused.blochsim_rfgr(M0, rf, gr, spd['loc'], Δf=spd['Δf'], γ_beff=spd['γ'], consts=consts)
        Mo.sum().backward()
    return dict(Mo=Mo.detach(), gM0=M0.grad, grf=rf.grad, ggr=gr.grad, **extra)


def exact(sp, pulse, consts_cpu, field_f32):
    nM = sp['M0'].shape[1]
    cc = C.constants_from(consts_cpu['γ2πdt'], consts_cpu['E1'], consts_cpu['E2'], consts_cpu['E1_1'],
                          N=1, nM=nM)
    Mo, gMi, grf, ggr = C.blochsim_rfgr_grad(sp['M0'], pulse['rf'], pulse['gr'], sp['loc'], Δf=sp['Δf'],
                                             γ_beff=sp['γ'], consts=cc, field_f32=field_f32)
    return dict(Mo=Mo, gM0=gMi, grf=grf, ggr=ggr), cc


def dist(a, e):
    return {k: rel_l2(a[k], e[k]) for k in ('Mo', 'gM0', 'grf', 'ggr')}


def subset_case(name, cfg, G, pulse, ref=None):
    idx, sp, _ = cases.big_subset(cfg, torch.float32, 4096)
    assert np.array_equal(idx.numpy(), G['idx'])
    cc_cpu = gconsts(G, 'cpu')
    t0 = time.perf_counter()
    ex32, cc = exact(sp, pulse, cc_cpu, True)
    ex64, _ = exact(sp, pulse, cc_cpu, False)
    rec = {'oracle_s': round(time.perf_counter() - t0, 2),
           'exact_f64field_vs_f32field': dist(ex64, ex32)}
    for mode in ('precise', 'fast'):
        with mrphy_amd.precision(mode), mrphy_amd.constants_on('cpu'):
            two = hip_grads(sp, pulse, gconsts(G), 'two')
            fus = hip_grads(sp, pulse, gconsts(G), 'fused')
        rec[f'hip_two_{mode}'] = dist(two, ex32)
        rec[f'hip_fused_{mode}'] = dist(fus, ex32)
        rec[f'hip_two_{mode}_vs_f64field'] = dist(two, ex64)
        rec[f'fused_vs_two_{mode}'] = {k: rel_l2(fus[k], two[k]) for k in ('gM0', 'grf', 'ggr')}
        if mode == 'precise':
            # stages of the two-kernel route: K3's grad_Beff against the exact adjoint over the SAME
            # fp32 Beff; the K0 adjoint's sums against fp64 sums of the SAME grad_Beff
            gMi_e, gB_e = C.blochsim_bwd(sp['M0'], two['beff'].cpu(), torch.ones_like(sp['M0']), consts=cc)
            gB = two['gBeff'].double().cpu()
            loc = sp['loc'].double()
            rec['stage_K3_gBeff_vs_exact'] = rel_l2(gB, gB_e)
            rec['stage_K3_gBeff_xyz_vs_exact'] = [rel_l2(gB[..., i], gB_e[..., i]) for i in range(3)]
            ggr64 = torch.einsum('nsk,nst->nkt', loc, gB[..., 2])
            grf64 = gB[..., :2].sum(1).permute(0, 2, 1)
            rec['stage_K0adj_sum_vs_f64sum'] = {'grf': rel_l2(two['grf'], grf64), 'ggr': rel_l2(two['ggr'], ggr64)}
            ggr_e = torch.einsum('nsk,nst->nkt', loc, gB_e[..., 2])
            rec['exact_gBeff_chain_vs_fused_oracle'] = rel_l2(ggr_e, ex32['ggr'])
            # how much cancellation is in the spin sums: |sum| / sum|.|
            rec['ggr_cancellation'] = float(ggr_e.abs().sum() / torch.einsum('nsk,nst->nkt', loc.abs(), gB_e[..., 2].abs()).sum())
            rec['norms'] = {k: float(ex32[k].norm()) for k in ex32}
    if ref is not None:
        rec['reference_golden'] = {k: rel_l2(ref[k], ex32[k]) for k in ref}
        rec['hip_two_precise_vs_reference'] = None
    out[name] = rec
    print(name, json.dumps(rec, indent=1), flush=True)


I = golden('interp_f32')
G4 = golden('big_cfg4_f32')
p5 = dict(rf=t(I['rf']), gr=t(I['gr']), dt=t(I['dt']))
subset_case('cfg5_subset_64c_x2048', 4, G4, p5,
            ref=dict(Mo=G4['Mo_sims'], grf=G4['grad_rf'], ggr=G4['grad_gr']))
G2 = golden('big_cfg2_f32')
_, _, p2 = cases.big_subset(2, torch.float32, 4096)
subset_case('cfg2_subset_128c_x4096', 2, G2, p2)

# whole config 5 (all 262144 spins), constants = the device's own
n, nT = 64, 2048
spd = synth.cube_spins(n, dtype=torch.float32, device=DEV, seed_M0=2004)
sp = {k: v.cpu() for k, v in spd.items()}
g_, E1_, E2_, E1m1_ = sims.relax_constants(spd['T1'], spd['T2'], spd['γ'], dev(p5['dt']), 4, DEV)
consts = {'γ2πdt': g_, 'E1': E1_, 'E2': E2_, 'E1_1': E1m1_}
t0 = time.perf_counter()
ex, cc = exact(sp, p5, {k: v.cpu() for k, v in consts.items()}, True)
rec = {'oracle_s': round(time.perf_counter() - t0, 2)}
for mode in ('precise', 'fast'):
    with mrphy_amd.precision(mode):
        rec[f'hip_two_{mode}'] = dist(hip_grads(sp, p5, consts, 'two'), ex)
        rec[f'hip_fused_{mode}'] = dist(hip_grads(sp, p5, consts, 'fused'), ex)
out['cfg5_whole_64c_x2048'] = rec
print('cfg5_whole', json.dumps(rec, indent=1), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
