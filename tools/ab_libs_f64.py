"""Same-box A/B of two builds of the library on the fp64 path (64^3 x 1024): K1, K1h + K3, fused K2 / K2b.
    python tools/ab_libs_f64.py LIB_A.so LIB_B.so      (one child process per library)"""
import os
import subprocess
import sys

if len(sys.argv) >= 3 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip() or r.stderr[-600:], flush=True)
    sys.exit(0)
assert sys.argv[1] == '--child' and len(sys.argv) == 3, 'usage: ab_libs_f64.py LIB_A.so LIB_B.so'
lib = sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib, beffective, sims, fused, synth  # noqa: E402
_lib.library_path = lambda: os.path.abspath(lib)
dev = torch.device('cuda', 0)
n, nT = 64, 1024
f64 = torch.float64
sp = synth.cube_spins(n, dtype=f64, device=dev, seed_M0=3)
p = synth.pulse(nT, dtype=f64, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
ss = n ** 3 * nT


def t_of(fn, reps=6):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts) // 2 - 1], out


with torch.no_grad():
    t0, beff = t_of(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']))
    t1, Mo = t_of(lambda: sims.blochsim(sp['M0'], beff, **kw))
    t2, Mf = t_of(lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw))
beff.requires_grad_(True)
Mi = sp['M0'].clone().requires_grad_(True)
t3, Mo2 = t_of(lambda: sims.blochsim(Mi, beff, **kw))
t4, g = t_of(lambda: torch.autograd.grad(Mo2, (Mi, beff), torch.ones_like(Mo2), retain_graph=True))
print(f'fp64 {n}^3x{nT}: K0 {t0:.3f} ms ({24 * ss / t0 / 1e9:.2f} TB/s)  K1 {t1:.3f} ms ({24 * ss / t1 / 1e9:.2f} TB/s)  '
      f'K1h {t3:.3f} ({48 * ss / t3 / 1e9:.2f})  K3 {t4:.3f} ({72 * ss / t4 / 1e9:.2f})  K2 {t2:.3f} ms ({ss / t2 / 1e6:.0f} G ss/s)  '
      f'|Mo| {float(Mo.norm()):.15e} |gB| {float(g[1].norm()):.15e} fused==two {bool((Mf == Mo).all())}')
