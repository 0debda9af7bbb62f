"""What does K0's write leave behind that slows the K1 that follows it?  (K1 on a block K0 has just written runs
at 0.55-0.65 of HBM peak; the same K1 on the same block a second time at 0.72-0.8.)  Sequences, each timed
per launch with HIP events, dev build (K0 variants through MRPHY_K0_VARIANT = order*1000 + rows/8*10 + nt):
    python tools/k0k1_state.py OUT.json"""
import json
import os
import statistics
import sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev  # noqa: E402
build_dev.use()
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth  # noqa: E402
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
res = []
for label, n, nM, nT in (('cfg1 64^3x1024', 64, 64 ** 3, 1024), ('shard 262144x4096', 128, 262144, 4096), ('128^3x1024', 128, 128 ** 3, 1024)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    alg = 12 * nM * nT + 36 * nM
    frac = lambda ms: round(alg / (ms * 1e-3) / 8e12, 3)  # noqa: E731
    with torch.no_grad():
        A = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
        B = torch.empty_like(A)
        flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB: 4 x the MALL

        def k0(out, variant='0'):
            os.environ['MRPHY_K0_VARIANT'] = variant
            beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=out)

        def k1(b):
            a_, b_ = ev(), ev()
            a_.record(); sims.blochsim(sp['M0'], b, **kw); b_.record()
            return a_, b_

        def seq(name, prep, nk1=3, reps=7):
            ts = [[] for _ in range(nk1)]
            tp = []
            for rep in range(reps + 1):
                p0, p1 = ev(), ev()
                p0.record(); prep(); p1.record()
                evs = [k1(A) for _ in range(nk1)]
                torch.cuda.synchronize()
                if rep:
                    tp.append(p0.elapsed_time(p1))
                    for i, (a_, b_) in enumerate(evs):
                        ts[i].append(a_.elapsed_time(b_))
            r = dict(size=label, seq=name, prep_ms=round(statistics.median(tp), 4),
                     K1_ms=[round(statistics.median(t), 4) for t in ts],
                     K1_frac=[frac(statistics.median(t)) for t in ts])
            print(json.dumps(r), flush=True); res.append(r)

        k0(A); k0(B)
        for v, nm in (('0', 'shipped'), ('1321', 'pinned 5/6-step batches')):
            os.environ['MRPHY_FWD_VARIANT'] = v
            for xcd in ('0', '1', '2'):
                os.environ['MRPHY_K1_XCD'] = xcd
                seq(f'[K1 {nm}, XCD-contiguous={xcd}] K0(A) -> K1 K1 K1', lambda: k0(A))
        os.environ['MRPHY_FWD_VARIANT'] = '1321'
        for kv, nm in (('161', 'spin-tile-fastest 128 rows nt'), ('21', 'spin-tile-fastest 16 rows nt'), ('1021', 'time-tile-fastest 16 rows nt')):
            for xcd in ('0', '1'):
                os.environ['MRPHY_K1_XCD'] = xcd
                seq(f'[K1 pinned, XCD-contiguous={xcd}] K0(A, {nm}) -> K1 K1 K1', lambda: k0(A, kv))
        os.environ['MRPHY_FWD_VARIANT'] = '0'; os.environ['MRPHY_K1_XCD'] = '0'
    del A, B, flush, sp
    torch.cuda.empty_cache()
os.environ['MRPHY_K0_VARIANT'] = '0'
json.dump({'device': torch.cuda.get_device_name(0), 'runs': res}, open(sys.argv[1], 'w'), indent=1)
