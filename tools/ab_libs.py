"""Same-box A/B of two builds of the library on the fused gradient path (VALU-bound kernels differ by
~10 % between boxes, so only numbers from one gpurun call compare).

    python tools/ab_libs.py LIB_A.so LIB_B.so      (runs itself once per library in a child process)
"""
import os
import subprocess
import sys

if len(sys.argv) == 3 and sys.argv[1] != '--child':
    for lib in sys.argv[1:3]:
        for rep in range(2):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], capture_output=True, text=True)
            print(os.path.basename(lib), f'run {rep}:', r.stdout.strip() or r.stderr[-400:], flush=True)
    sys.exit(0)

assert sys.argv[1] == '--child' and len(sys.argv) == 3, 'usage: ab_libs.py LIB_A.so LIB_B.so'
lib = sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib, fused, synth  # noqa: E402
_lib.library_path = lambda: os.path.abspath(lib)
dev = torch.device('cuda', 0)
out = []
for n, nT in ((64, 2048), (128, 1024)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    tf, tb = [], []
    for it in range(12):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        Mo = fused.blochsim_rfgr(sp['M0'], rf, gr, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
        e[1].record()
        Mo.sum().backward()
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
    med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
    out.append(f'{n}^3x{nT}: fwd {med(tf):.3f} bwd {med(tb):.3f} ms  |grf| {float(rf.grad.double().norm()):.9e}')
print('  '.join(out))
