"""Is the 0.64-vs-0.76 spread of K1h / K3 a property of one-generation grids, or of where the
buffers they WRITE landed?  For each size, several trials with freshly allocated history /
grad_Beff blocks (torch.cuda.empty_cache() re-rolls the driver's placement): the plain fill_ rate of
the very block each kernel writes, next to the kernel's time.

    python tools/placement_vs_size.py OUT.json [trials]
"""
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402

dev = torch.device('cuda', 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def t_of(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = ev(), ev()
        torch.cuda.synchronize()
        a.record(); out = fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, out


def fill_rate(numel):
    r"""TB/s of a streaming write of a fresh fp32 block of `numel` elements -- which the caching
    allocator serves from the block the kernel's output just vacated."""
    blk = torch.empty(numel, dtype=torch.float32, device=dev)
    ms, _ = t_of(lambda: blk.fill_(0.5))
    p = blk.data_ptr()
    del blk
    return numel * 4 / ms / 1e9, p


res = []
for n, nT in ((64, 2048), (128, 1024)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    ss = n ** 3 * nT
    numel = ss * 3
    for tr in range(trials):
        torch.cuda.empty_cache()
        # a spacer of varying size shifts where the next blocks land
        spacer = torch.empty((tr * 1536 + 1) << 20, dtype=torch.uint8, device=dev)
        with torch.no_grad():
            beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        beff.requires_grad_(True)
        Mi = sp['M0'].clone().requires_grad_(True)
        tf, Mo = t_of(lambda: sims.blochsim(Mi, beff, **kw), reps=1)        # allocates the history
        tf2 = []
        for _ in range(3):                       # the same history block is NOT reused by a new call:
            del Mo                               # free it first, then the allocator hands it out again
            t, Mo = t_of(lambda: sims.blochsim(Mi, beff, **kw), reps=1)
            tf2.append(t)
        tb = []
        for _ in range(3):
            t, g = t_of(lambda: torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo), retain_graph=True), reps=1)
            tb.append(t)
            gptr = g[1].data_ptr()
            del g
        fr_g, pg = fill_rate(numel)              # the block grad_Beff just vacated
        del Mo
        fr_h, ph = fill_rate(numel)              # the block the history just vacated
        r = dict(cube=n, nT=nT, trial=tr, K1h_ms=round(min(tf2), 4), K1h_frac=round(24 * ss / min(tf2) / 8e9, 3),
                 hist_block_fill_TBps=round(fr_h, 3), K3_ms=round(min(tb), 4),
                 K3_frac=round(36 * ss / min(tb) / 8e9, 3), gBeff_block_fill_TBps=round(fr_g, 3),
                 gBeff_block_reused=bool(pg == gptr))
        print(json.dumps(r), flush=True)
        res.append(r)
        del beff, Mi, spacer
json.dump({'note': 'per trial: time of K1h (writes the history) and K3 (writes grad_Beff) and the plain fill_ '
                   'rate of the very blocks they wrote', 'runs': res}, open(sys.argv[1], 'w'), indent=1)
