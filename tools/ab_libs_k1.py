"""Same-box A/B of two builds of the library on K1 (blochsim over a materialised Beff), each in a child
process, twice:   python tools/ab_libs_k1.py LIB_A.so LIB_B.so"""
import os
import subprocess
import sys

if len(sys.argv) == 3 and sys.argv[1] != '--child':
    for rep in range(2):
        for lib in sys.argv[1:3]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], capture_output=True, text=True)
            print(os.path.basename(lib), f'run {rep}:', r.stdout.strip() or r.stderr[-400:], flush=True)
    sys.exit(0)
lib = sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib, beffective, sims, synth  # noqa: E402
_lib.library_path = lambda: os.path.abspath(lib)
dev = torch.device('cuda', 0)
out = []
for n, nT in ((64, 1024), (64, 2048), (128, 4096)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        for mode in ('precise', 'fast'):
            ts = []
            with mrphy_amd.precision(mode):
                for it in range(10):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); a.record()
                    Mo = sims.blochsim(sp['M0'], beff, **kw)
                    b.record(); torch.cuda.synchronize()
                    if it >= 2:
                        ts.append(a.elapsed_time(b))
            out.append(f'{n}^3x{nT} {mode}: {sorted(ts)[len(ts) // 2]:.3f} (min {min(ts):.3f}) |Mo| {float(Mo.double().norm()):.9e}')
    del beff
print('  '.join(out))
