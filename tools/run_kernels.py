"""Launch selected kernels a few times on the synthetic cube -- the program that goes after
``rocprofv3 ... --`` for kernel-stats and PMC passes.

    python tools/run_kernels.py WHAT CUBE NT [REPS [SPINS]]        (SPINS: only the first SPINS spins of the cube -- a rank's shard)
WHAT: k2 (fused forward, precise then fast) | fwd (K0 + K1) | grad (K0, K1h, K3, K0 adjoint)
      | gradws (as grad, through a placement-probed workspace.GradWorkspace: its probe launches are the fp64-constant
      instances `prec_f64` / run first) | gradfused (K2 with checkpoints + K2b); a trailing 64 (fwd64, grad64, gradws64,
      k264) runs the same in fp64
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT]
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, fused, synth  # noqa: E402

what, n, nT = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device('cuda', 0)
dt_ = torch.float32
if what.endswith('64'):
    what, dt_ = what[:-2], torch.float64
spins = int(sys.argv[5]) if len(sys.argv) > 5 else None
sp = synth.cube_spins(n, None if spins is None else torch.arange(spins), dtype=dt_, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=dt_, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
if what == 'k2':
    for mode in ('precise', 'fast'):
        with mrphy_amd.precision(mode), torch.no_grad():
            for _ in range(reps):
                fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
            torch.cuda.synchronize()
elif what == 'fwd':
    with torch.no_grad():
        for _ in range(reps):
            beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
            sims.blochsim(sp['M0'], beff, **kw)
            del beff
        torch.cuda.synchronize()
elif what == 'grad':
    for _ in range(reps):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        sims.blochsim(sp['M0'], beff, **kw).sum().backward()
        del beff
    torch.cuda.synchronize()
elif what == 'gradws':
    from mrphy_amd import workspace
    ws = workspace.GradWorkspace((1, sp['M0'].shape[1], nT, 3), dt_, dev)
    print('workspace', ws.report)
    for _ in range(reps):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=ws.beff)
        sims.blochsim(sp['M0'], beff, workspace=ws, **kw).sum().backward()
        del beff
    torch.cuda.synchronize()
elif what == 'gradfused':
    for _ in range(reps):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw).sum().backward()
    torch.cuda.synchronize()
else:
    sys.exit(f'unknown {what}')
print('done', what, n, nT, reps)
