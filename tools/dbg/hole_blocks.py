"""Can a block be MADE of scattered physical chunks with plain allocations?  profiles/r05_arena_marks_*: a window of a big
allocation is fast for K1h when it spans a boundary between two physical chunks of the allocation.  Here: allocate 2k
chunks of C MiB, free every other one and return them to the driver (holes), then allocate the 6-GiB block -- if the driver
builds it from the holes it consists of k scattered chunks -- and time K1h / K3 on it; then free the other chunks and
allocate a second block out of THEIR holes.  Controls: plain blocks before and after.

    python tools/dbg/hole_blocks.py OUT.json
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
numel = nM * nT * 3
L = numel * 4
MiB = 1 << 20
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
other = torch.empty(numel, dtype=torch.float32, device=dev)
Mi = torch.zeros((1, nM, 3), device=dev)
Mi[..., 2] = 1
T = torch.ones((), device=dev)
kw = dict(T1=T, T2=T * 0.07)
gMo = torch.ones_like(Mi)


def measure(t):
    tH = timed(lambda: sims.blochsim(Mi, beff, workspace=_Pair(t, other), **kw))
    Mo = sims.blochsim(Mi, beff, workspace=_Pair(other, t), **kw)
    tG = timed(lambda: torch.autograd.grad(Mo, beff, gMo, retain_graph=True))
    return round(tH, 3), round(tG, 3)


rows = []


def rec(**k):
    print(json.dumps(k), flush=True)
    rows.append(k)


for i in range(3):
    b = torch.empty(numel, dtype=torch.float32, device=dev)
    rec(kind='plain block', i=i, K1h_K3_ms=measure(b))
    del b
    torch.cuda.empty_cache()
for C in (256, 512, 1024, 2048):
    k = -(-L // (C * MiB)) + 1
    chunks = [torch.empty(C * MiB // 4, dtype=torch.float32, device=dev) for _ in range(2 * k)]
    odd = chunks[1::2]
    del chunks
    torch.cuda.empty_cache()                       # the even chunks go back to the driver: k holes of C MiB
    a = torch.empty(numel, dtype=torch.float32, device=dev)
    rec(kind=f'block allocated into {k} holes of {C} MiB', K1h_K3_ms=measure(a))
    del odd
    torch.cuda.empty_cache()                       # now the odd chunks are holes
    b = torch.empty(numel, dtype=torch.float32, device=dev)
    rec(kind=f'second block, into the other {k} holes of {C} MiB', K1h_K3_ms=measure(b))
    rec(kind='  (first block again)', K1h_K3_ms=measure(a))
    del a, b
    torch.cuda.empty_cache()
for i in range(3):
    b = torch.empty(numel, dtype=torch.float32, device=dev)
    rec(kind='plain block afterwards', i=i, K1h_K3_ms=measure(b))
    del b
    torch.cuda.empty_cache()
json.dump({'cube': n, 'nT': nT, 'rows': rows}, open(sys.argv[1], 'w'), indent=1)
