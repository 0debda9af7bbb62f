"""K1 / K1h / K2 / K2b timing and accuracy, 'fast' vs 'precise' step (MRPHY_FWD_VARIANT honoured).

    python tools/precision_sweep.py [cube] [nT]
"""
import os
import sys
import time

import torch

sys.path[:0] = ['.', 'oracle']
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev()
        a.record()
        out = f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sum(ts) / len(ts), out


# exact reference on a subset (oracle/bloch_c.c, fp64 arithmetic on the same fp32 field + constants)
import bloch_c as C  # noqa: E402
idx = synth.subset_indices(n, 16384, seed=5)
spc, pc = synth.cube_spins(n, idx, dtype=torch.float32), synth.pulse(nT, dtype=torch.float32)
with mrphy_amd.constants_on('cpu'):
    g, E1, E2, E1_1 = sims.relax_constants(spc['T1'], spc['T2'], spc['γ'], pc['dt'], 4, dev)
consts = dict(γ2πdt=g, E1=E1, E1_1=E1_1, E2=E2)
spd = {k: v.to(dev) for k, v in spc.items()}
with torch.no_grad():
    b_sub = beffective.rfgr2beff(p['rf'], p['gr'], spd['loc'], Δf=spd['Δf'], γ=spd['γ'])
want = C.blochsim(spc['M0'], b_sub.cpu(), consts=C.constants_from(g, E1, E2, E1_1, N=1, nM=idx.numel())) \
    if hasattr(C, 'blochsim') else None

for mode in ('fast', 'precise'):
    with mrphy_amd.precision(mode), torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        k1 = timeit(lambda: sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt']))
        del beff
        k2 = timeit(lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'],
                                                γ_beff=sp['γ'], T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt']))
        eq = bool((k1[2] == k2[2]).all())
        with mrphy_amd.constants_on('cpu'):
            Ms = sims.blochsim_consts(spd['M0'], b_sub, **consts)
        err = float((Ms.double().cpu() - want).norm() / want.norm()) if want is not None else float('nan')
    gb = 12 * n ** 3 * nT / 1e9
    print(f'{mode:8s} K1 {k1[0]:7.3f} ms ({gb / k1[0]:.3f} TB/s)  K2 {k2[0]:7.3f} ms '
          f'({n ** 3 * nT / k2[0] / 1e9:.3f} T ss/s)  K2==K1: {eq}  rel-L2 vs exact (16k spins): {err:.3e}',
          flush=True)
