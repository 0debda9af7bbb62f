"""CPU emulation (numpy, one rounding per emulated fp32 operation) of the kernels' fp32 step, 'fast'
and 'precise' (csrc/bloch_math.hpp), on a seeded subset of the headline workload, against fp64
arithmetic on the same fp32 field and constants.  This is how the precise step was designed: it
shows which roundings matter and reproduces the GPU's error figures to three digits.

    python tools/precision_emul.py [nT=4096] [spins=1024]

Findings on 128^3 x 4096 (phi up to 2.6 rad per step), relative L2 from exact arithmetic:
    variant                                         seeded M0     M0 = z
    fast (fp32 Horner S, C; 3-4 roundings/comp.)     2.0e-5       2.5e-5
    S, C correctly rounded only                      1.0e-5       2.4e-5
    compensated update only                          1.8e-5       4.0e-6
    precise (both)                                   4.8e-6       1.7e-6
The fp32 Horner error of S (1.4e-7) and C (3.9e-8) enters the state multiplied by phi and phi^2;
the z component of the plain update (p = s E1; p - (E1 - 1)) loses T1 recovery systematically.
"""
import sys

import numpy as np
import torch

sys.path[:0] = ['.', 'oracle']
from mrphy_amd import synth  # noqa: E402
import bloch_oracle as O  # noqa: E402

f32, f64 = np.float32, np.float64
# csrc/bloch_math.hpp: rot_coeffs_poly (fp32), rot_coeffs_poly_precise (fp64 evaluation)
S32 = [1.361460111e-10, -2.472925686e-08, 2.753590024e-06, -1.984053670e-04, 8.333321661e-03, -1.666666567e-01, 1.0]
C32 = [-1.773506675e-09, 2.721793635e-07, -2.478447095e-05, 1.388849691e-03, -4.166663438e-02, 0.5]
S64 = [-6.61101325761093948e-13, 1.58967818563764548e-10, -2.50387235830435598e-08, 2.75567047781280402e-06,
       -1.98412544903332901e-04, 8.33333314478239967e-03, -1.66666666578391104e-01, 9.99999999993218314e-01]
C64 = [9.92981381380750462e-12, -2.06726098894046996e-09, 2.75437529915325051e-07, -2.48011224326891956e-05,
       1.38888812875101854e-03, -4.16666662001028004e-02, 4.99999999953231744e-01]


def fma(a, b, c):          # fp32 FMA: exact product and sum in fp64, one rounding
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def mul(a, b):
    return (a * b).astype(f32)


def horner32(co, x):
    s = np.full_like(x, f32(co[0]))
    for c in co[1:]:
        s = fma(s, x, np.full_like(x, f32(c)))
    return s


def horner64(co, x):
    s = np.full_like(x, co[0])
    for c in co[1:]:
        s = s * x + c
    return s


def cross(a, b):
    return (fma(a[1], b[2], -mul(a[2], b[1])), fma(a[2], b[0], -mul(a[0], b[2])), fma(a[0], b[1], -mul(a[1], b[0])))


def setup(n, nT, count, seed_M0):
    idx = synth.subset_indices(n, count, seed=7)
    sp = synth.cube_spins(n, idx, dtype=torch.float32, seed_M0=seed_M0)
    p = synth.pulse(nT, dtype=torch.float32)
    beff = O.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])[0].numpy()
    g = (2 * np.pi * sp['γ'] * p['dt']).to(torch.float32).numpy().reshape(-1)[0]
    E1, E2 = torch.exp(-p['dt'] / sp['T1'])[0].numpy(), torch.exp(-p['dt'] / sp['T2'])[0].numpy()
    return beff, g, E1, E2, (E1 - f32(1)).astype(f32), sp['M0'][0].numpy()


def exact(beff, g, E1, E2, E1m1, M):
    m = [M[:, i].astype(f64) for i in range(3)]
    for t in range(beff.shape[1]):
        b = [beff[:, t, i].astype(f64) * f64(g) for i in range(3)]
        x = b[0] ** 2 + b[1] ** 2 + b[2] ** 2
        ph = np.sqrt(x)
        S = np.where(ph > 1e-8, np.sin(ph) / np.maximum(ph, 1e-300), 1 - x / 6)
        C = np.where(ph > 1e-4, (1 - np.cos(ph)) / np.maximum(x, 1e-300), 0.5 - x / 24)
        w = (b[1] * m[2] - b[2] * m[1], b[2] * m[0] - b[0] * m[2], b[0] * m[1] - b[1] * m[0])
        v = (b[1] * w[2] - b[2] * w[1], b[2] * w[0] - b[0] * w[2], b[0] * w[1] - b[1] * w[0])
        m = [m[i] - S * w[i] + C * v[i] for i in range(3)]
        m = [m[0] * E2, m[1] * E2, m[2] * E1 - E1m1.astype(f64)]
    return np.stack(m, -1)


def run(data, sc_precise, upd_precise):
    beff, g, E1, E2, E1m1, M = data
    m = [M[:, i].copy() for i in range(3)]
    G = np.full(M.shape[0], g, f32)
    E = [E2, E2, E1]
    D = [(e - f32(1)).astype(f32) for e in E]
    for t in range(beff.shape[1]):
        b = [mul(beff[:, t, i], G) for i in range(3)]
        x = fma(b[2], b[2], fma(b[1], b[1], mul(b[0], b[0])))
        if sc_precise:
            S, C = horner64(S64, x.astype(f64)).astype(f32), horner64(C64, x.astype(f64)).astype(f32)
        else:
            S, C = horner32(S32, x), horner32(C32, x)
        w = cross(b, m)
        v = cross(b, w)
        new = []
        for i in range(3):
            a = fma(-S, w[i], m[i])
            s = fma(C, v[i], a)
            if upd_precise:                   # update_precise<RELAX, OFFSET>
                es = fma(C, v[i], (a - s).astype(f32))
                r = fma(s, D[i], -E1m1) if i == 2 else mul(s, D[i])
                new.append((s + (es + r).astype(f32)).astype(f32))
            else:                             # rot_apply, fast
                p = mul(s, E[i])
                new.append((p - E1m1).astype(f32) if i == 2 else p)
        m = new
    return np.stack([q.astype(f64) for q in m], -1)


if __name__ == '__main__':
    nT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    for seed_M0, label in ((11, 'seeded M0'), (None, 'M0 = z')):
        data = setup(128, nT, count, seed_M0)
        ex = exact(*data)
        rel = lambda got: np.linalg.norm(got - ex) / np.linalg.norm(ex)  # noqa: E731
        print(f'128^3 x {nT}, {count} spins, {label}:')
        for name, scp, upp in (('fast', False, False), ('S, C correctly rounded only', True, False),
                               ('compensated update only', False, True), ('precise', True, True)):
            print(f'    {name:30s} {rel(run(data, scp, upp)):.3e}', flush=True)
