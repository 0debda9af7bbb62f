"""beff2ab forward + gradient to beff: the fused adjoint against the composed route it replaced (four
differentiable blochsim calls, columns of [I | 0]).   python tools/ab_timing.py [cube] [nT]"""
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
E1, E2 = torch.exp(-p['dt'] / sp['T1']), torch.exp(-p['dt'] / sp['T2'])
with torch.no_grad():
    beff0 = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
NNd = beff0.shape[:-2]


def fused():
    b = beff0.detach().requires_grad_(True)
    A, B = beffective.beff2ab(b, E1=E1, E2=E2, γ=sp['γ'], dt=p['dt'])
    (A.sum() + B.sum()).backward()
    return A, B, b.grad


def composed():
    b = beff0.detach().requires_grad_(True)
    g = 2 * torch.pi * sp['γ'] * p['dt']
    zero = torch.zeros_like(E1)
    cols = []
    for j in range(3):
        e = torch.zeros(NNd + (3,), device=dev)
        e[..., j] = 1
        cols.append(sims.blochsim_consts(e, b, γ2πdt=g, E1=E1, E1_1=zero, E2=E2))
    B = sims.blochsim_consts(torch.zeros(NNd + (3,), device=dev), b, γ2πdt=g, E1=E1, E1_1=E1 - 1, E2=E2)
    A = torch.stack(cols, dim=-1)
    (A.sum() + B.sum()).backward()
    return A, B, b.grad


res = {}
for name, f in (('fused', fused), ('composed', composed)):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = ev(), ev(); a.record(); out = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    res[name] = (min(ts), out)
    print(f'{name:9s} forward+backward {min(ts):8.3f} ms  ({n ** 3 * nT / min(ts) / 1e6:7.1f} G spin-steps/s)', flush=True)
Af, Bf, gf = res['fused'][1]
Ac, Bc, gc = res['composed'][1]
rel = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
print(f'A equal {torch.equal(Af, Ac)}  B equal {torch.equal(Bf, Bc)}  grad rel-L2 {rel(gf, gc):.2e}')
