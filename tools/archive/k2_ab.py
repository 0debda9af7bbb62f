"""The fused forward kernel K2 across dev builds (MRPHY_DEV_TAG, one child process per library) and team on/off:
span from the per-wave stamps and the HIP-event time of back-to-back launches.
    python tools/k2_ab.py OUT.json TAG[,TAG...]        ('' = the plain dev build)"""
import json, os, statistics, subprocess, sys
if len(sys.argv) == 3:
    out = {}
    for tag in sys.argv[2].split(','):
        env = dict(os.environ, MRPHY_DEV_TAG=tag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        print(f'== build "{tag}"'); print(r.stdout.strip() or r.stderr[-800:], flush=True)
        try:
            out[tag] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
        except Exception:
            pass
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
    sys.exit(0)
import numpy as np
import torch
sys.path[:0] = ['.', 'tools']
import build_dev
lib = build_dev.use()
import mrphy_amd
from mrphy_amd import fused, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for n, nT in ((64, 1024), (64, 2048), (128, 1024), (128, 4096)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    f = lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
    tiles = n ** 3 // 64
    stamps = torch.zeros((tiles, 4), dtype=torch.int64, device=dev)
    with torch.no_grad():
        for mode in ('precise', 'fast'):
            for team in ('0', '1'):
                if team == '1' and tiles > 4096:
                    continue
                os.environ['MRPHY_K2_TEAM'] = team
                with mrphy_amd.precision(mode):
                    for _ in range(3):
                        Mo = f()
                    torch.cuda.synchronize()
                    a, b = ev(), ev()
                    a.record()
                    for _ in range(10):
                        f()
                    b.record(); torch.cuda.synchronize()
                    ms = a.elapsed_time(b) / 10
                    stamps.zero_(); lib.mrphy_dev_set_stamps(stamps.data_ptr(), tiles)
                    f(); torch.cuda.synchronize()
                    lib.mrphy_dev_set_stamps(None, 0)
                    s = stamps.cpu().numpy(); s = s[s[:, 1] > 0]
                    span = (s[:, 1].max() - s[:, 0].min()) * 1e-2
                print(json.dumps(dict(cube=n, nT=nT, mode=mode, team=team, ms_back_to_back=round(ms, 4), span_us=round(float(span), 1),
                                      norm=float(Mo.double().norm()))), flush=True)
