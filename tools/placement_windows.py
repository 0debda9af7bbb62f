"""Where inside ONE big allocation are the fast and the slow windows for the gradient route's write kernels?

A single arena of ARENA_GB; K1h writes its history (and K3 its grad_Beff) into a window of the block size that slides
through it in steps of STEP_MB.  If fast / slow is a property of coarse physical regions the profile is piecewise
constant; if it depends on how the window's start is aligned it changes from step to step.

    python tools/placement_windows.py OUT.json [cube nT arena_GB step_MB [fwd]]

With a sixth argument `fwd` the window is the Beff block instead: K0 (rfgr2beff, write-only, under both store policies)
and K1 (blochsim without history, read-only) on each window.
"""
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import sims  # noqa: E402
from mrphy_amd.workspace import _Pair  # noqa: E402

dev = torch.device('cuda', 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nT = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
arena_gb = float(sys.argv[4]) if len(sys.argv) > 4 else 48
step_mb = float(sys.argv[5]) if len(sys.argv) > 5 else 512
nM = n ** 3
numel = nM * nT * 3
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def timed(fn, reps=2):
    fn()
    best = 1e9
    for _ in range(reps):
        a, b = ev(), ev()
        a.record(); fn(); b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


fwd = len(sys.argv) > 6 and sys.argv[6] == 'fwd'
if fwd:
    from mrphy_amd import beffective, synth
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    arena = torch.empty(int(arena_gb * (1 << 30)) // 4, dtype=torch.float32, device=dev)
    step = int(step_mb * (1 << 20)) // 4
    rows, off = [], 0
    with torch.no_grad():
        while off + numel <= arena.numel():
            win = arena[off:off + numel].view(1, nM, nT, 3)
            t0 = {st: timed(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=win,
                                                         store=st)) for st in ('nt', 'sc1nt')}
            t1 = timed(lambda: sims.blochsim(sp['M0'], win, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt']))
            rows.append(dict(offset_MB=off * 4 / (1 << 20), K0_nt_ms=round(t0['nt'], 4), K0_sc1nt_ms=round(t0['sc1nt'], 4),
                             K1_ms=round(t1, 4)))
            print(json.dumps(rows[-1]), flush=True)
            off += step
    json.dump({'cube': n, 'nT': nT, 'arena_GB': arena_gb, 'step_MB': step_mb, 'arena_ptr': hex(arena.data_ptr()),
               'block_bytes': numel * 4, 'windows': rows}, open(sys.argv[1], 'w'), indent=1)
    sys.exit(0)

field = torch.empty(numel, dtype=torch.float32, device=dev)
beff = field.view(1, nM, nT, 3)
beff.uniform_(-2.0, 2.0)
beff.requires_grad_(True)
other = torch.empty(numel, dtype=torch.float32, device=dev)
arena = torch.empty(int(arena_gb * (1 << 30)) // 4, dtype=torch.float32, device=dev)
Mi = torch.zeros((1, nM, 3), device=dev)
Mi[..., 2] = 1
T = torch.ones((), device=dev)
kw = dict(T1=T, T2=T * 0.07)
gMo = torch.ones_like(Mi)
step = int(step_mb * (1 << 20)) // 4
rows = []
off = 0
while off + numel <= arena.numel():
    win = arena[off:off + numel]
    tH = timed(lambda: sims.blochsim(Mi, beff, workspace=_Pair(win, other), **kw))
    Mo = sims.blochsim(Mi, beff, workspace=_Pair(other, win), **kw)
    tG = timed(lambda: torch.autograd.grad(Mo, beff, gMo, retain_graph=True))
    del Mo
    tF = timed(lambda: win.fill_(0.5))
    rows.append(dict(offset_MB=off * 4 / (1 << 20), K1h_ms=round(tH, 4), K3_ms=round(tG, 4), fill_ms=round(tF, 4)))
    print(json.dumps(rows[-1]), flush=True)
    off += step
json.dump({'cube': n, 'nT': nT, 'arena_GB': arena_gb, 'step_MB': step_mb, 'arena_ptr': hex(arena.data_ptr()),
           'block_bytes': numel * 4, 'windows': rows}, open(sys.argv[1], 'w'), indent=1)
