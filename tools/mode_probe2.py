"""Does the K3 / K1h rate follow the PLACEMENT of the big buffers?  Within one process: allocate
(Beff, history, grad_Beff), time K1h + K3, free everything back to the driver, allocate a spacer of
a different size, repeat.   python tools/mode_probe2.py [tag]"""
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
tag = sys.argv[1] if len(sys.argv) > 1 else ''
n, nT = 128, 1024
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
gb = 12 * n ** 3 * nT / 1e9
line = []
for trial, spacer_gib in enumerate((0, 5, 0, 17, 2, 40)):
    torch.cuda.empty_cache()
    spacer = torch.empty(spacer_gib << 30, dtype=torch.uint8, device=dev) if spacer_gib else None
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    beff.requires_grad_(True)
    Mi = sp['M0'].clone().requires_grad_(True)
    tf, tb = [], []
    for it in range(4):
        a, b, c = ev(), ev(), ev()
        a.record(); Mo = sims.blochsim(Mi, beff, **kw); b.record()
        g = torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo)); c.record()
        torch.cuda.synchronize()
        if it:
            tf.append(a.elapsed_time(b)); tb.append(b.elapsed_time(c))
        del g, Mo
    line.append(f'[{spacer_gib:2d}G K1h {min(tf):5.2f} K3 {min(tb):5.2f}]')
    del beff, Mi, spacer
print(f'{tag:>3} ' + ' '.join(line), flush=True)
