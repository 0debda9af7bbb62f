"""Where is the fast window of a big allocation, as a function of its size, and does a second big allocation in the same
process have one too?  (profiles/r05_placement_windows_*: 64-120-GiB allocations are fast for K1h / K3 / K0 only across
their 32-GiB mark.)  Several arenas are allocated one after the other and KEPT; K1h's history window (64^3 x 2048:
6 GiB) slides through each in 1-GiB steps.

    python tools/dbg/arena_marks.py OUT.json [sizes in GiB ...]
"""
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import sims  # noqa: E402
from mrphy_amd.workspace import _Pair  # noqa: E402

dev = torch.device('cuda', 0)
sizes = [float(x) for x in sys.argv[2:]] or [36, 40, 48, 40, 34]
n, nT = 64, 2048
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


field = torch.empty(numel, dtype=torch.float32, device=dev)
beff = field.view(1, nM, nT, 3)
beff.uniform_(-2.0, 2.0)
beff.requires_grad_(True)
dummy = torch.empty(16, device=dev)
Mi = torch.zeros((1, nM, 3), device=dev)
Mi[..., 2] = 1
T = torch.ones((), device=dev)
kw = dict(T1=T, T2=T * 0.07)
GiB = 1 << 30
out, keep = [], []
for gb in sizes:
    arena = torch.empty(int(gb * GiB) // 4, dtype=torch.float32, device=dev)
    keep.append(arena)
    prof = []
    off = 0
    while off + numel <= arena.numel():
        win = arena[off:off + numel]
        prof.append(round(timed(lambda: sims.blochsim(Mi, beff, workspace=_Pair(win, dummy), **kw)), 3))
        off += GiB // 4
    best = min(range(len(prof)), key=prof.__getitem__)
    r = dict(arena_GiB=gb, ptr=hex(arena.data_ptr()), K1h_ms_by_offset_GiB=prof, fastest_offset_GiB=best, fastest_ms=prof[best],
             slowest_ms=max(prof))
    print(json.dumps(r), flush=True)
    out.append(r)
json.dump({'cube': n, 'nT': nT, 'block_GiB': numel * 4 / GiB, 'arenas': out}, open(sys.argv[1], 'w'), indent=1)
