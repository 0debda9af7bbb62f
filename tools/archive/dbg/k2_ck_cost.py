"""What writing checkpoints costs the fused forward: K2 under no_grad (plain build) against K2 with requires_grad (checkpoint build),
1 / 2 / 8 coils, 64^3 x 1024 and x 2048, back-to-back launches."""
import sys, statistics
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import fused, synth
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
def t_of(fn, reps=5, inner=6):
    ts = []
    for i in range(reps + 1):
        a, b = ev(), ev(); a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        if i:
            ts.append(a.elapsed_time(b) / inner)
    return statistics.median(ts)
g = torch.Generator(device='cpu').manual_seed(5)
for nT in (1024, 2048):
    sp = synth.cube_spins(64, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    for nC in (1, 2, 8):
        rf, b1 = p['rf'], None
        if nC > 1:
            b1 = (torch.randn((1, 64 ** 3, 2, nC), generator=g) / nC).to(dev)
            rf = (p['rf'].unsqueeze(-1) * torch.linspace(0.5, 1.5, nC, device=dev)).contiguous()
        f = lambda r: fused.blochsim_rfgr(sp['M0'], r, p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], b1Map=b1, **kw)
        with torch.no_grad():
            t_plain = t_of(lambda: f(rf))
        rg = rf.clone().requires_grad_(True)
        t_ck = t_of(lambda: f(rg))
        print(f'64^3 x {nT}, {nC} coil(s): plain {t_plain:.4f} ms, with checkpoints {t_ck:.4f} ms ({100 * (t_ck / t_plain - 1):+.1f} %)', flush=True)
