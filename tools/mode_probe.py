"""One process = one sample of the box's "mode": torch fill_ / sum (plain streaming write / read), K0, K1,
K1h, K3 at 128^3 x 1024, each the min of 4 after a warm-up, plus pointers of the big buffers.
Run it several times in one gpurun call and compare processes.   python tools/mode_probe.py [tag]"""
import os
import subprocess
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
tag = sys.argv[1] if len(sys.argv) > 1 else ''
n, nT = 128, 1024
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def t_min(f, reps=4):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev(); a.record(); r = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)); del r
    return min(ts)


def smi():
    try:
        o = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.split(':', 2)[-1].strip() for ln in o.splitlines() if 'GPU[0]' in ln and any(k in ln for k in ('sclk', 'mclk', 'fclk', 'socclk'))]
        return ' | '.join(keep)
    except Exception as e:
        return f'rocm-smi failed: {e}'


sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
gb = 12 * n ** 3 * nT / 1e9
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    fill = t_min(lambda: beff.fill_(1.0))
    rd = t_min(lambda: beff.sum())
    k0 = t_min(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']))
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    k1 = t_min(lambda: sims.blochsim(sp['M0'], beff, **kw))
clk_mid = smi()
beff.requires_grad_(True)
Mi = sp['M0'].clone().requires_grad_(True)
tf, tb = [], []
for it in range(5):
    a, b, c = ev(), ev(), ev()
    a.record(); Mo = sims.blochsim(Mi, beff, **kw); b.record()
    g = torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo)); c.record()
    torch.cuda.synchronize()
    if it:
        tf.append(a.elapsed_time(b)); tb.append(b.elapsed_time(c))
    ptr_g = g[1].data_ptr()
    del g, Mo
print(f'{tag:>3} fill {gb / fill:5.2f}  sum {gb / rd:5.2f}  K0 {gb / k0:5.2f}  K1 {gb / k1:5.2f}  K1h {2 * gb / min(tf):5.2f}  '
      f'K3 {3 * gb / min(tb):5.2f} TB/s | K0 {k0:6.3f} K1 {k1:6.3f} K1h {min(tf):6.3f} K3 {min(tb):6.3f} ms | '
      f'beff 0x{beff.data_ptr():x} gB 0x{ptr_g:x} | {clk_mid}', flush=True)
