"""fp64 polynomials for the rotation coefficients on [0, pi^2]: interpolation at Chebyshev nodes in
50-digit arithmetic (near-minimax), monomial coefficients rounded to double, error of the fp64
Horner evaluation against the exact functions.   python tools/fit_poly64.py"""
import math

import mpmath as mp
import numpy as np

mp.mp.dps = 60
A = mp.pi ** 2


def f(kind, x):
    x = mp.mpf(x)
    if x == 0:
        return {'S': mp.mpf(1), 'C': mp.mpf(1) / 2, 'dS': -mp.mpf(1) / 6, 'dC': -mp.mpf(1) / 24}[kind]
    r = mp.sqrt(x)
    S, C = mp.sin(r) / r, (1 - mp.cos(r)) / x
    if kind == 'S':
        return S
    if kind == 'C':
        return C
    if kind == 'dS':                       # (cos r - S) / (2x)
        return (mp.cos(r) - S) / (2 * x)
    return (S / 2 - C) / x                 # dC


def fit(kind, deg):
    n = deg + 1
    xs = [A / 2 * (1 - mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
    V = mp.matrix(n, n)
    for i, x in enumerate(xs):
        for j in range(n):
            V[i, j] = x ** j
    y = mp.matrix([f(kind, x) for x in xs])
    c = mp.lu_solve(V, y)
    return [c[j] for j in range(n)]


def horner64(coef, x):
    acc = np.full_like(x, np.float64(coef[-1]))
    for c in coef[-2::-1]:
        acc = acc * x + np.float64(c)
    return acc


if __name__ == '__main__':
    grid = np.linspace(0, float(A), 20001)
    for kind, deg in (('S', 12), ('C', 12), ('dS', 12), ('dC', 12), ('S', 13), ('C', 13)):
        c = fit(kind, deg)
        exact = np.array([float(f(kind, x)) for x in grid[::10]])
        approx_mp = np.array([float(sum(cj * mp.mpf(x) ** j for j, cj in enumerate(c))) for x in grid[::10]])
        e_fit = np.max(np.abs(approx_mp - exact))
        e64 = np.max(np.abs(horner64([float(v) for v in c], grid[::10]) - exact))
        print(f'{kind:>2} degree {deg}: fit err {e_fit:.2e}  fp64 Horner max abs err {e64:.2e}')
        print('    ' + ', '.join(mp.nstr(v, 20) for v in c[::-1]))
