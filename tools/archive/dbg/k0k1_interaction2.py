"""Freshly written, or another block?  K1 at 64^3 x 1024 / the 1/8 shard: (a) alone on block A, (b) after an
IN-PLACE rewrite of block A (torch copy_ from a second tensor: same physical block, fresh data), (c) reading the
ping-pong blocks K0 returns (a new allocation per call)."""
import json, statistics, sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
for n, nT, shard in ((64, 1024, 1), (128, 4096, 8)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    if shard > 1:
        m = n ** 3 // shard
        sp = {k: (v[:, :m].contiguous() if torch.is_tensor(v) and v.ndim >= 2 and v.shape[1] == n ** 3 else v) for k, v in sp.items()}
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    k0 = lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    res = {}
    with torch.no_grad():
        A = k0(); src = k0()
        ptrs = set()
        for mode in ('alone_A', 'A_rewritten_in_place', 'alone_A_again', 'fresh_blocks_from_K0'):
            ts = []
            cur = A
            for it in range(14):
                if mode == 'A_rewritten_in_place': A.copy_(src)
                if mode == 'fresh_blocks_from_K0': cur = k0(); ptrs.add(cur.data_ptr())
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); Mo = sims.blochsim(sp['M0'], cur, **kw); b.record(); torch.cuda.synchronize()
                if it >= 2: ts.append(round(a.elapsed_time(b), 3))
            res[mode] = dict(median=round(statistics.median(ts), 4), all=ts)
    print(json.dumps(dict(cube=n, nT=nT, shard_of=shard, distinct_fresh_blocks=len(ptrs), K1_ms=res)), flush=True)
    del A, src, cur
