"""Is the K0-write / K1-read rate a property of the BLOCK?  Allocates `nblk` separate Beff blocks (one hipMalloc
each through torch's caching allocator), then, interleaved over `reps` rounds, times K0 writing each block and
K1 reading it (HIP events), and finally the bench's own pattern (a fresh `rfgr2beff` allocation per step).
    python tools/block_probe.py OUT.json [nblk] [reps]        (MRPHY_PROBE_SIZES=0,1 selects sizes)"""
import json
import os
import statistics
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402
dev = torch.device('cuda', 0)
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
SIZES = [('cfg1 64^3x1024', 64, 64 ** 3, 1024), ('shard 262144x4096', 128, 262144, 4096)]
if os.environ.get('MRPHY_PROBE_SIZES'):
    SIZES = [SIZES[int(i)] for i in os.environ['MRPHY_PROBE_SIZES'].split(',')]
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
out = []
for label, n, nM, nT in SIZES:
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT
    with torch.no_grad():
        blocks = [torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev) for _ in range(nblk)]
        k0t = [[] for _ in blocks]; k1t = [[] for _ in blocks]
        for rep in range(reps + 1):
            for i, b in enumerate(blocks):
                e = [ev() for _ in range(3)]
                e[0].record()
                beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=b)
                e[1].record()
                sims.blochsim(sp['M0'], b, **kw)
                e[2].record(); torch.cuda.synchronize()
                if rep:
                    k0t[i].append(e[0].elapsed_time(e[1])); k1t[i].append(e[1].elapsed_time(e[2]))
        rows = [dict(block=i, ptr=hex(b.data_ptr()), K0_ms=round(statistics.median(k0t[i]), 4),
                     K1_ms=round(statistics.median(k1t[i]), 4), K1_min=round(min(k1t[i]), 4),
                     K0_TBps=round(alg / statistics.median(k0t[i]) / 1e9, 2),
                     K1_frac=round((alg + 36 * nM) / statistics.median(k1t[i]) / 1e9 / 8000, 3))
                for i, b in enumerate(blocks)]
        for r in rows:
            print(label, json.dumps(r), flush=True)
        # the bench's pattern: a fresh allocation per step, the previous one released after K1
        del blocks
        torch.cuda.empty_cache()
        k0b, k1b, ptrs = [], [], []
        for rep in range(2 * reps + 2):
            e = [ev() for _ in range(3)]
            e[0].record()
            beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
            e[1].record()
            sims.blochsim(sp['M0'], beff, **kw)
            e[2].record(); torch.cuda.synchronize()
            ptrs.append(hex(beff.data_ptr()))
            if rep >= 2:
                k0b.append(round(e[0].elapsed_time(e[1]), 4)); k1b.append(round(e[1].elapsed_time(e[2]), 4))
            del beff
        bench_like = dict(K0_ms=k0b, K1_ms=k1b, ptrs=ptrs[2:],
                          K1_frac_median=round((alg + 36 * nM) / statistics.median(k1b) / 1e9 / 8000, 3))
        print(label, 'bench-like', json.dumps(bench_like), flush=True)
    out.append(dict(size=label, spins=nM, nT=nT, blocks=rows, bench_like=bench_like))
    del sp
    torch.cuda.empty_cache()
json.dump({'device': torch.cuda.get_device_name(0), 'runs': out}, open(sys.argv[1], 'w'), indent=1)
