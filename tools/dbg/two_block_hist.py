"""Is a block fast when its two halves are SEPARATE allocations written at the same time?

profiles/r05_placement_windows_*: K1h (which writes the history) is fast only when its history block lies across the
32-GiB mark of a big allocation -- i.e., if the runtime builds big allocations from 32-GiB pieces, when the two halves
of the block are in two different pieces.  Here the 64^3 x 2048 forward-with-history runs as TWO half-size launches on
two streams, concurrently, each writing its own history block:
  one launch, one 6.4-GB block                                    (the product's way; several blocks)
  two launches, the two halves of ONE 6.4-GB block                (control: same memory, two launches)
  two launches, two separately allocated 3.2-GB blocks (i, j)     (neighbours and far apart)

    python tools/dbg/two_block_hist.py OUT.json
"""
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import sims  # noqa: E402
from mrphy_amd.workspace import _Pair  # noqa: E402

dev = torch.device('cuda', 0)
n, nT = 64, 2048
nM = n ** 3
half = nM // 2
numel, hnumel = nM * nT * 3, half * nT * 3
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
field = torch.empty(numel, dtype=torch.float32, device=dev)
field.uniform_(-2.0, 2.0)
beff = field.view(1, nM, nT, 3).requires_grad_(True)
b_lo = field[:hnumel].view(1, half, nT, 3).requires_grad_(True)
b_hi = field[hnumel:].view(1, half, nT, 3).requires_grad_(True)
Mi = torch.zeros((1, nM, 3), device=dev)
Mi[..., 2] = 1
Mi_h = Mi[:, :half].contiguous()
T = torch.ones((), device=dev)
kw = dict(T1=T, T2=T * 0.07)
dummy = torch.empty(16, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def one(block):
    best = 1e9
    for it in range(3):
        a, b = ev(), ev()
        a.record(); sims.blochsim(Mi, beff, workspace=_Pair(block, dummy), **kw); b.record()
        b.synchronize()
        if it:
            best = min(best, a.elapsed_time(b))
    return round(best, 4)


def two(hA, hB):
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize()
        a, eA, eB = ev(), ev(), ev()
        a.record()
        sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sA):
            sims.blochsim(Mi_h, b_lo, workspace=_Pair(hA, dummy), **kw)
            eA.record()
        with torch.cuda.stream(sB):
            sims.blochsim(Mi_h, b_hi, workspace=_Pair(hB, dummy), **kw)
            eB.record()
        eA.synchronize(); eB.synchronize()
        if it:
            best = min(best, max(a.elapsed_time(eA), a.elapsed_time(eB)))
    return round(best, 4)


rows = []


def rec(**k):
    print(json.dumps(k), flush=True)
    rows.append(k)


big = [torch.empty(numel, dtype=torch.float32, device=dev) for _ in range(4)]
for i, b in enumerate(big):
    rec(kind='one launch, one 6.4-GB block', block=i, ms=one(b))
for i, b in enumerate(big):
    rec(kind='two launches, halves of ONE 6.4-GB block', block=i, ms=two(b[:hnumel], b[hnumel:]))
small = [torch.empty(hnumel, dtype=torch.float32, device=dev) for _ in range(10)]
for i, j in ((0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (0, 9), (1, 8), (2, 7), (3, 6), (0, 5), (4, 9)):
    rec(kind='two launches, two separate 3.2-GB blocks', i=i, j=j, ms=two(small[i], small[j]))
# one half-size launch alone on each small block (what a single stream to one BO does)
for i in range(10):
    best = 1e9
    for it in range(3):
        a, b = ev(), ev()
        a.record(); sims.blochsim(Mi_h, b_lo, workspace=_Pair(small[i], dummy), **kw); b.record()
        b.synchronize()
        if it:
            best = min(best, a.elapsed_time(b))
    rec(kind='ONE half-size launch alone, 3.2-GB block', i=i, ms=round(best, 4))
# mixed: one half in a big block, the other in a small one
for i in range(4):
    rec(kind='two launches: first half of big block i, small block i', i=i, ms=two(big[i][:hnumel], small[i]))
json.dump({'cube': n, 'nT': nT, 'rows': rows}, open(sys.argv[1], 'w'), indent=1)
