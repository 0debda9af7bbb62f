"""The VALU-bound kernels for several builds of the library, one child process each, twice (alternating):
    python tools/ab_libs_valu.py LIB_A.so LIB_B.so ...
fp32 precise: fused forward K2 (64^3 x 2048, 128^3 x 1024), fused forward + adjoint K2 + K2b (64^3 x 2048), 8 transmit coils
fused forward + adjoint (64^3 x 1024); fp64: K2 and K2 + K2b (64^3 x 1024).  Norms printed to compare bits."""
import os, subprocess, sys
if len(sys.argv) >= 3 and sys.argv[1] != '--child':
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], capture_output=True, text=True)
            print(os.path.basename(lib), f'run {rep}:', r.stdout.strip() or r.stderr[-400:], flush=True)
    sys.exit(0)
lib = sys.argv[2]
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')]
import statistics, torch
import mrphy_amd
from mrphy_amd import _lib, fused, synth
_lib.library_path = lambda: os.path.abspath(lib)
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)


def t_of(fn, reps=6, inner=6):
    ts = []
    for i in range(reps + 1):
        a, b = ev(), ev()
        a.record()
        for _ in range(inner):
            out = fn()
        b.record(); torch.cuda.synchronize()
        if i >= 1:
            ts.append(a.elapsed_time(b) / inner)
    return statistics.median(ts), out


out = []
for label, dt, n, nT, nC in (('f32 64^3x2048', torch.float32, 64, 2048, 1), ('f32 128^3x1024', torch.float32, 128, 1024, 1),
                             ('f32 2coils 64^3x1024', torch.float32, 64, 1024, 2), ('f32 8coils 64^3x1024', torch.float32, 64, 1024, 8), ('f64 64^3x1024', torch.float64, 64, 1024, 1)):
    sp = synth.cube_spins(n, dtype=dt, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=dt, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    rf, b1 = p['rf'], None
    if nC > 1:
        g = torch.Generator(device='cpu').manual_seed(5)
        b1 = (torch.randn((1, n ** 3, 2, nC), generator=g, dtype=dt) / nC).to(dev)
        rf = (p['rf'].unsqueeze(-1) * torch.linspace(0.5, 1.5, nC, dtype=dt, device=dev)).contiguous()
    f = lambda rf_=rf, gr_=p['gr']: fused.blochsim_rfgr(sp['M0'], rf_, gr_, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], b1Map=b1, **kw)
    with torch.no_grad():
        t_f, Mo = t_of(f)
    s = f'{label}: K2 {t_f:.4f}'
    if n == 64:
        def fb():
            r_, g_ = rf.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
            f(r_, g_).sum().backward()
            return r_.grad
        t_fb, gr = t_of(fb, 5, 3)
        r_, g_ = rf.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        Mo_ = f(r_, g_)
        t_b, _ = t_of(lambda: torch.autograd.grad(Mo_, (r_, g_), torch.ones_like(Mo_), retain_graph=True), 5, 3)
        bits = lambda x: int(x.contiguous().view(torch.int32 if x.dtype == torch.float32 else torch.int64).to(torch.int64).sum())  # noqa: E731
        r2_, g2_ = rf.clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        M2_ = sp['M0'].clone().requires_grad_(True)
        fused.blochsim_rfgr(M2_, r2_, g2_, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], b1Map=b1, **kw).sum().backward()
        s += (f' fwd+bwd {t_fb:.4f} bwd {t_b:.4f} |grf| {float(gr.double().norm()):.12e} '
              f'bits(grf, ggr, gM0) {bits(r2_.grad)} {bits(g2_.grad)} {bits(M2_.grad)}')
        del Mo_
    out.append(s + f' |Mo| {float(Mo.double().norm()):.12e}')
    del sp
# the materialised route in fp32 (HBM-bound): K1 on a resident block, K1 with history + K3   (MRPHY_AB_NO_MAT=1 skips it)
from mrphy_amd import beffective, sims
for n, nT in (() if os.environ.get('MRPHY_AB_NO_MAT') else ((64, 2048), (128, 1024))):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        t1, Mo = t_of(lambda: sims.blochsim(sp['M0'], beff, **kw), 5, 4)
    beff.requires_grad_(True)
    Mi = sp['M0'].clone().requires_grad_(True)
    t1h, Mo2 = t_of(lambda: sims.blochsim(Mi, beff, **kw), 4, 2)
    t3, g = t_of(lambda: torch.autograd.grad(Mo2, (Mi, beff), torch.ones_like(Mo2), retain_graph=True), 4, 2)
    ss = n ** 3 * nT
    out.append(f'f32 {n}^3x{nT}: K1 {t1:.3f} ({12 * ss / t1 / 8e9:.3f}) K1h {t1h:.3f} ({24 * ss / t1h / 8e9:.3f}) K3 {t3:.3f} ({36 * ss / t3 / 8e9:.3f}) '
               f'|gB| {float(g[1].double().norm()):.12e}')
    del beff, Mo2, g, sp
print('   '.join(out))
