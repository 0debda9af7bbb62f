"""Cheaper evaluations of S(x) = sin(phi)/phi, C(x) = (1 - cos(phi))/phi^2 for the PRECISE fp32 step,
emulated on the CPU (numpy, one rounding per emulated fp32 operation: tools/precision_emul.py) on a seeded subset of
the headline workload, against fp64 arithmetic on the same fp32 field and constants -- BEFORE anything is built.

    python tools/precise_sc_variants.py [nT=4096] [spins=1024]

Issue slots of the S, C evaluation per step (an fp64 FMA or a conversion = 2, an fp32 operation = 1); the shipped
precise step spends 32 of its 74 slots per wave-step here:
    v0  S, C: degree-7 / 6 fits evaluated in fp64, rounded once (shipped)              13 FMA64 + 3 cvt   = 32
    v1  S as v0; C: the fast step's degree-5 fp32 Horner                               7 FMA64 + 2 cvt + 5 = 23
    v1b S: degree-6 fit in fp64; C fp32                                                6 FMA64 + 2 cvt + 5 = 21
    v3  S: fp32 Horner, last two levels compensated (error-free product and sum); C fp32           20 + 5 = 25
Why C needs no help: its Horner error is dominated by the final rounding (3e-8), the next level adds 1.8e-8 at
x = pi^2; S's level-1 rounding (0.5 ulp of 1/6) is amplified by x to 1e-7.
"""
import sys

import numpy as np

sys.path[:0] = ['.', 'oracle', 'tools']
import precision_emul as E  # noqa: E402

f32, f64 = np.float32, np.float64
fma, mul, horner32, horner64 = E.fma, E.mul, E.horner32, E.horner64
add = lambda a, b: (a + b).astype(f32)  # noqa: E731
sub = lambda a, b: (a - b).astype(f32)  # noqa: E731
# degree-6 near-minimax fit of S on [0, pi^2] (tools/fit_poly.py --fp64 --deg 6): filled in by fit() below
S64_6 = None


def fit_S6():
    r"""Degree-6 least-squares fit at Chebyshev nodes (near-minimax), in fp64."""
    k = np.arange(4000)
    x = 0.5 * np.pi ** 2 * (1 + np.cos(np.pi * (k + 0.5) / 4000))
    ph = np.sqrt(x)
    y = np.sin(ph) / ph
    co = np.polynomial.chebyshev.chebfit(2 * x / np.pi ** 2 - 1, y, 6)
    p = np.polynomial.chebyshev.cheb2poly(co)                   # in u = 2x/pi^2 - 1
    # to a polynomial in x
    P = np.polynomial.Polynomial(p)(np.polynomial.Polynomial([-1, 2 / np.pi ** 2]))
    c = P.coef[::-1]
    err = np.abs(np.polyval(c, x) - y).max()
    return list(c), err


def S_comp(x):
    r"""v3: fp32 Horner down to h2 = a2 + x (...), then levels 1 and 0 with error-free transformations."""
    co = [f32(c) for c in E.S64]
    a1, a1lo = f32(E.S64[6]), f32(E.S64[6] - f64(f32(E.S64[6])))
    h2 = horner32(E.S64[:6], x)
    p = mul(x, h2); ep = fma(x, h2, -p)
    v = add(np.full_like(x, a1), p); es = sub(p, sub(v, np.full_like(x, a1)))
    vlo = add(add(ep, es), np.full_like(x, a1lo))
    q = mul(x, v); eq = fma(x, v, -q)
    one = np.ones_like(x)
    S0 = add(one, q); eS = sub(q, sub(S0, one))
    return add(S0, add(eS, fma(x, vlo, eq)))


def make(variant):
    def sc(x):
        if variant == 'v0':
            return horner64(E.S64, x.astype(f64)).astype(f32), horner64(E.C64, x.astype(f64)).astype(f32)
        if variant == 'v1':
            return horner64(E.S64, x.astype(f64)).astype(f32), horner32(E.C32, x)
        if variant == 'v1b':
            return horner64(S64_6, x.astype(f64)).astype(f32), horner32(E.C32, x)
        if variant == 'v1c':                    # C with the degree-6 coefficients, fp32 Horner
            return horner64(E.S64, x.astype(f64)).astype(f32), horner32(E.C64, x)
        if variant == 'v3':
            return S_comp(x), horner32(E.C32, x)
        if variant == 'fastSC':
            return horner32(E.S32, x), horner32(E.C32, x)
        raise ValueError(variant)
    return sc


def run(data, sc):
    beff, g, E1, E2, E1m1, M = data
    m = [M[:, i].copy() for i in range(3)]
    G = np.full(M.shape[0], g, f32)
    D = [(e - f32(1)).astype(f32) for e in (E2, E2, E1)]
    for t in range(beff.shape[1]):
        b = [mul(beff[:, t, i], G) for i in range(3)]
        x = fma(b[2], b[2], fma(b[1], b[1], mul(b[0], b[0])))
        S, C = sc(x)
        w = E.cross(b, m)
        v = E.cross(b, w)
        new = []
        for i in range(3):
            a = fma(-S, w[i], m[i])
            s = fma(C, v[i], a)
            es = fma(C, v[i], (a - s).astype(f32))
            r = fma(s, D[i], -E1m1) if i == 2 else mul(s, D[i])
            new.append((s + (es + r).astype(f32)).astype(f32))
        m = new
    return np.stack([q.astype(f64) for q in m], -1)


if __name__ == '__main__':
    nT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    S64_6, e6 = fit_S6()
    print(f'degree-6 fit of S: max approximation error {e6:.2e}')
    xs = np.linspace(0, np.pi ** 2, 200001).astype(f32)
    ph = np.sqrt(xs.astype(f64))
    Sx = np.where(ph > 1e-6, np.sin(ph) / np.maximum(ph, 1e-300), 1.0)
    Cx = np.where(ph > 1e-3, (1 - np.cos(ph)) / np.maximum(xs.astype(f64), 1e-300), 0.5 - xs.astype(f64) / 24)
    for v in ('v0', 'v1', 'v1b', 'v1c', 'v3', 'fastSC'):
        S, C = make(v)(xs)
        print(f'  {v:7s} max |S - exact| {np.abs(S - Sx).max():.2e}   max |C - exact| {np.abs(C - Cx).max():.2e}'
              f'   max x|C - exact| {(xs * np.abs(C - Cx)).max():.2e}')
    for seed_M0, label in ((11, 'seeded M0'), (None, 'M0 = z')):
        data = E.setup(128, nT, count, seed_M0)
        ex = E.exact(*data)
        rel = lambda got: np.linalg.norm(got - ex) / np.linalg.norm(ex)  # noqa: E731
        print(f'128^3 x {nT}, {count} spins, {label}:')
        for v in ('v0', 'v1', 'v1b', 'v1c', 'v3', 'fastSC'):
            got = run(data, make(v))
            print(f'    {v:8s} rel-L2 {rel(got):.3e}   max abs {np.abs(got - ex).max():.3e}', flush=True)
