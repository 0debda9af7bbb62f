"""Near-minimax polynomial fits (Remez exchange, in fp64/longdouble) for the rotation coefficients

    S(x) = sin(sqrt x)/sqrt x,   C(x) = (1 - cos(sqrt x))/x,   S'(x), C'(x)      on  [0, pi^2]

used by csrc/bloch_math.hpp.  Prints the float coefficients (highest degree first) and the
maximum absolute error of the fp32 Horner evaluation against the fp64 series.

    python tools/fit_poly.py
"""
import math

import numpy as np

A = math.pi ** 2


def series(x, kind):
    """fp64 (longdouble) reference by Taylor series, 40 terms (converges fast on [0, pi^2])."""
    x = np.asarray(x, dtype=np.longdouble)
    out = np.zeros_like(x)
    for k in range(39, -1, -1):
        if kind == 'S':
            c = (-1) ** k / math.factorial(2 * k + 1)
        elif kind == 'C':
            c = (-1) ** k / math.factorial(2 * k + 2)
        elif kind == 'dS':      # d/dx: sum_{k>=1} (-1)^k k x^(k-1)/(2k+1)!
            c = (-1) ** (k + 1) * (k + 1) / math.factorial(2 * k + 3)
        elif kind == 'dC':
            c = (-1) ** (k + 1) * (k + 1) / math.factorial(2 * k + 4)
        out = out * x + np.longdouble(c)
    return out


def remez(kind, deg, iters=30):
    n = deg + 2
    # Chebyshev extrema on [0, A] as the initial reference
    xs = (A / 2) * (1 - np.cos(np.pi * np.arange(n) / (n - 1)))
    grid = np.linspace(0, A, 20001)
    for _ in range(iters):
        V = np.vander(xs, deg + 1, increasing=True).astype(np.float64)
        M = np.hstack([V, ((-1.0) ** np.arange(n))[:, None]])
        sol = np.linalg.solve(M, series(xs, kind).astype(np.float64))
        coef = sol[:-1]
        err = np.polyval(coef[::-1], grid) - series(grid, kind).astype(np.float64)
        # new reference: extrema of the error between sign changes
        idx = [0]
        for i in range(1, len(grid)):
            if np.sign(err[i]) != np.sign(err[idx[-1]]) and err[i] != 0:
                idx.append(i)
            elif abs(err[i]) > abs(err[idx[-1]]):
                idx[-1] = i
        if len(idx) != n:
            break
        new = grid[idx]
        if np.max(np.abs(new - xs)) < 1e-9:
            xs = new
            break
        xs = new
    return coef, np.max(np.abs(err))


def horner32(coef, x):
    x = x.astype(np.float32)
    acc = np.full_like(x, np.float32(coef[-1]))
    for c in coef[-2::-1]:
        # emulate fmaf: product and sum in fp64, one rounding to fp32
        acc = (acc.astype(np.float64) * x.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)
    return acc


if __name__ == '__main__':
    grid = np.linspace(0, A, 400001)
    for kind, deg in (('S', 6), ('C', 5), ('S', 5), ('C', 6), ('dS', 5), ('dC', 5), ('dS', 6), ('dC', 4)):
        coef, e = remez(kind, deg)
        e32 = np.max(np.abs(horner32(coef, grid).astype(np.float64) - series(grid, kind).astype(np.float64)))
        print(f'{kind:>2} degree {deg}: minimax err {e:.2e}, fp32 Horner max abs err {e32:.2e}')
        print('    ' + ', '.join(f'{np.float32(c):.9e}f' for c in coef[::-1]))
