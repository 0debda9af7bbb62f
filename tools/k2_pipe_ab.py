"""The fused forward kernel K2: the shipped build against the software-pipelined one-coil builds (dev knob
MRPHY_K2_PIPE = chunk length 4 / 8), across dev libraries (MRPHY_DEV_TAG; one child process per library):
HIP-event time of back-to-back launches, results compared bit for bit with the shipped build's.
    python tools/k2_pipe_ab.py OUT.json TAG[,TAG...]        ('' = the plain dev build)"""
import json, os, subprocess, sys
if len(sys.argv) == 3:
    out = {}
    for tag in sys.argv[2].split(','):
        env = dict(os.environ, MRPHY_DEV_TAG=tag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        print(f'== build "{tag}"'); print(r.stdout.strip() or r.stderr[-800:], flush=True)
        out[tag] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
    sys.exit(0)
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
    with torch.no_grad():
        for mode in ('precise', 'fast'):
            ref = None
            for pipe in os.environ.get('K2_PIPES', '0,4,8,0,4,8').split(','):
                os.environ['MRPHY_K2_PIPE'] = pipe
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
                if ref is None:
                    ref = Mo.clone()
                print(json.dumps(dict(cube=n, nT=nT, mode=mode, pipe=pipe, ms=round(ms, 4), G_ss_per_s=round(n ** 3 * nT / ms / 1e6, 1),
                                      same_bits=bool(torch.equal(Mo, ref)))), flush=True)
