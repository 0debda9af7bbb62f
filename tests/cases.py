r"""Closed-form inputs of every parity case.  Shared by ``tests/golden/make_golden.py`` (which
feeds them to the REFERENCE in the build container) and by the tests (which feed them to the
oracle and to the HIP path), so that only outputs need to be stored as golden vectors.

Case ``ref3``/``ref512`` reproduce the inputs of the reference's own tests
(``tests/test_slowsims.py:33-62``, ``tests/test_sims.py:36-61``); the others cover what the
reference's tests do not.
"""
from math import pi as π
import os
import sys

import torch
from torch import tensor

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

γH_val, dt0_val = 4257.6, 4e-6


def ref_pulse(nT: int, dtype, *, gr_y: float = 0.0):
    r"""Pulse of test_slowsims.py:54-61 / test_sims.py:55-61: rf = 10[cos,sin](2πt/nT) G with
    a trailing coil dim, gr = [1, gr_y, 10·atan(t - nT/2)/π] G/cm."""
    t = torch.arange(0, nT, dtype=dtype).reshape(1, 1, nT)
    rf = 10 * torch.cat([torch.cos(t / nT * 2 * π), torch.sin(t / nT * 2 * π)], 1)[..., None]
    gr = torch.cat([torch.ones((1, 1, nT), dtype=dtype), torch.full((1, 1, nT), gr_y, dtype=dtype),
                    10 * torch.atan(t - round(nT / 2)) / π], 1)
    return rf, gr


def ref_case(nM: int, dtype, *, nT: int = 512, seed: int = None):
    r"""The reference tests' spin line: loc_x = loc_y = linspace(-1,1,nM), loc_z = 1,
    Δf = -loc_x·γ (cancels gr_x = 1 G/cm), b1Map = [1, 0], T1 = 1 s, T2 = 40 ms.
    ``seed=None``: M0 = eye(3) (needs nM = 3, test_slowsims.py:35); else uniform [0,1)
    (test_sims.py:36, which is unseeded there)."""
    γ, dt = tensor(γH_val, dtype=dtype), tensor(dt0_val, dtype=dtype)
    if seed is None:
        assert nM == 3
        M0 = torch.eye(3, dtype=dtype)[None]
    else:
        gen = torch.Generator(device='cpu').manual_seed(seed)
        M0 = torch.rand((1, nM, 3), generator=gen, dtype=torch.float64).to(dtype)
    lx = torch.linspace(-1., 1., steps=nM, dtype=dtype).reshape(1, nM)
    loc = torch.stack([lx, lx.clone(), torch.ones((1, nM), dtype=dtype)], 2)
    rf, gr = ref_pulse(nT, dtype)
    return dict(M0=M0, loc=loc, Δf=-lx * γ, b1Map=tensor([1., 0.], dtype=dtype).reshape(1, 1, 2, 1),
                rf=rf, gr=gr, T1=tensor([[1.]], dtype=dtype), T2=tensor([[4e-2]], dtype=dtype),
                γ=γ, dt=dt)


def rfgr_variants(dtype, seed: int = 7):
    r"""F3: small rfgr2beff inputs covering every branch: N = 2, nM = 7, nT = 12, nC = 4."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    rnd = lambda *s: (torch.rand(s, generator=gen, dtype=torch.float64) * 2 - 1).to(dtype)  # noqa
    N, nM, nT, nC = 2, 7, 12, 4
    base = dict(gr=rnd(N, 3, nT), loc=rnd(N, nM, 3) * 5)
    γs = tensor(γH_val, dtype=dtype)
    γm = (γH_val * (1 + 0.1 * rnd(N, nM))).to(dtype)
    return {
        'plain':        dict(base, rf=rnd(N, 2, nT)),
        'df_scalar_γ':  dict(base, rf=rnd(N, 2, nT), Δf=rnd(N, nM) * 100, γ=γs),
        'df_map_γ':     dict(base, rf=rnd(N, 2, nT), Δf=rnd(N, nM) * 100, γ=γm),
        'coil1_dim':    dict(base, rf=rnd(N, 2, nT, 1), b1Map=rnd(N, nM, 2, 1)),
        'coil1_nodim':  dict(base, rf=rnd(N, 2, nT), b1Map=rnd(N, nM, 2)),
        'ptx4':         dict(base, rf=rnd(N, 2, nT, nC), b1Map=rnd(N, nM, 2, nC),
                             Δf=rnd(N, nM) * 100, γ=γm),
        'ptx4_nomap':   dict(base, rf=rnd(N, 2, nT, nC)),
        'batch1_pulse': dict(gr=rnd(1, 3, nT), loc=rnd(N, nM, 3) * 5, rf=rnd(1, 2, nT),
                             Δf=rnd(1, nM) * 100, γ=γs),
        'odd_nT':       dict(gr=rnd(N, 3, 13), loc=rnd(N, nM, 3) * 5, rf=rnd(N, 2, 13, 2),
                             b1Map=rnd(N, nM, 2, 2)),
    }


def bcast_variants(dtype, seed: int = 11):
    r"""F5: the broadcast zoo of sims.blochsim: N = 2, nM = 5, nT = 40; T1/T2/γ as 0-dim,
    (1,1), (N,1), (1,nM), (N,nM); dt as (), (1,), (N,)."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64)  # noqa: E731
    N, nM, nT = 2, 5, 40
    M0 = rnd(N, nM, 3).to(dtype)
    Beff = ((rnd(N, nM, nT, 3) * 2 - 1) * 8).to(dtype)
    Beff[0, 1] = 0                    # a spin that never rotates (phi = 0 every step)
    Beff[1, 2, 5:9] = 0               # zero-field steps in the middle of a pulse
    t1 = lambda *s: (0.5 + rnd(*s)).to(dtype)         # noqa: E731
    t2 = lambda *s: (0.02 + 0.1 * rnd(*s)).to(dtype)  # noqa: E731
    gm = lambda *s: (γH_val * (0.9 + 0.2 * rnd(*s))).to(dtype)  # noqa: E731
    dt = tensor(dt0_val, dtype=dtype)
    v = {
        'scalar':   dict(T1=t1().reshape(()), T2=t2().reshape(()), γ=gm().reshape(()), dt=dt),
        'one_one':  dict(T1=t1(1, 1), T2=t2(1, 1), γ=gm(1, 1), dt=dt.reshape(1)),
        'per_spin': dict(T1=t1(N, nM), T2=t2(N, nM), γ=gm(N, nM), dt=dt.reshape(1)),
        'mixed':    dict(T1=t1(N, 1), T2=t2(N, 1), γ=gm(1, nM), dt=dt.reshape(1)),  # T1,T2 must share a shape (sims.py:76 cat)
        'dt_batch': dict(T1=t1(N, nM), T2=t2(N, nM), γ=gm(1, 1),
                         dt=(dt0_val * (1 + rnd(N))).to(dtype)),
        'norelax':  dict(T1=None, T2=None, γ=gm(N, nM), dt=dt.reshape(1)),
        'expanded': dict(T1=t1(N, 1).expand(N, nM), T2=t2(1, 1).expand(N, nM),
                         γ=gm(1, 1).expand(N, nM), dt=dt.reshape(1)),   # stride-0 views (mobjs)
    }
    return M0, Beff, v


def onestep_case(dtype, seed: int = 13):
    r"""F4: blochsim_1step in/out, including the all-zero-field branch."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64)  # noqa: E731
    N, nM = 2, 9
    M = rnd(N, nM, 3).to(dtype)
    b = ((rnd(N, nM, 3) * 2 - 1) * 10).to(dtype)
    b[0, 3] = 0
    dt = tensor(dt0_val, dtype=dtype)
    T1, T2 = (0.5 + rnd(N, nM)).to(dtype), (0.02 + 0.1 * rnd(N, nM)).to(dtype)
    E1, E2 = torch.exp(-dt / T1), torch.exp(-dt / T2)
    g = 2 * π * tensor(γH_val, dtype=dtype) * dt
    return dict(M=M, b=b, E1=E1, E1_1=E1 - 1, E2=E2, γ2πdt=g)


def big_subset(cfg: int, dtype=torch.float32, count: int = 4096, *, coarse: bool = False):
    r"""F7: a seeded ``count``-spin subset of a BASELINE.json config's synthetic cube
    (spins are independent, so the subset's results equal the full run's rows)."""
    import mrphy_amd
    from mrphy_amd import synth
    c = synth.CONFIGS[cfg]
    idx = synth.subset_indices(c['n'], count, seed=1000 + cfg)
    spins = synth.cube_spins(c['n'], idx, dtype=dtype, seed_M0=2000 + cfg)
    if cfg == 4 and coarse:
        pulse = synth.pulse(c['nT'] // 2, dtype=dtype, dt=8e-6)
    else:
        pulse = synth.pulse(c['nT'], dtype=dtype)
    return idx, spins, pulse


def reference_constants(T1, T2, γ, dt, ndim: int):
    r"""γ2πdt, E1, E1-1, E2 formed exactly as the reference forms them (sims.py:62,74-76 after the
    right-padding of sims.py:309-313), on THIS machine's CPU.  ``make_golden.py`` stores them
    next to the reference's outputs: they are per-spin constants applied nT times, and exp() is
    not bit-reproducible across CPUs/GPUs, so a golden comparison must use the very constants
    the reference run used."""
    pad = lambda x: None if x is None else x.reshape(tuple(x.shape) + (ndim - x.ndim) * (1,))  # noqa
    γ, dt, T1, T2 = pad(γ), pad(dt), pad(T1), pad(T2)
    out = {'γ2πdt': 2 * π * γ * dt}
    if T1 is not None:
        E1, E2 = torch.exp(-dt / T1), torch.exp(-dt / T2)
        out.update(E1=E1, E1_1=E1 - 1, E2=E2)
    return out


def freeprec_variants(dtype, seed: int = 17):
    r"""free-precession cases: the reference's known answer (test_slowsims.py:100-121) and a small
    zoo of broadcast forms (N = 2, nM = 6)."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64)  # noqa: E731
    half = tensor([[0.5]], dtype=dtype)
    dur = tensor(0.5, dtype=dtype)
    Tk = -dur / torch.log(half)                                    # E1 = E2 = 0.5
    known = dict(M=torch.eye(3, dtype=dtype)[None], dur=dur, T1=Tk.reshape(()), T2=Tk.reshape(()),
                 Δf=tensor([[1 / 4 / 0.5, -1 / 4 / 0.5, 1]], dtype=dtype))
    N, nM = 2, 6
    M = rnd(N, nM, 3).to(dtype)
    t1, t2 = (0.5 + rnd(N, nM)).to(dtype), (0.02 + 0.1 * rnd(N, nM)).to(dtype)
    df = ((rnd(N, nM) * 2 - 1) * 300).to(dtype)
    return {
        'known':     known,
        'full':      dict(M=M, dur=tensor(3e-3, dtype=dtype), T1=t1, T2=t2, Δf=df),
        'dur_batch': dict(M=M, dur=tensor([1e-3, 4e-3], dtype=dtype), T1=t1, T2=t2, Δf=df),
        'scalars':   dict(M=M, dur=tensor([2e-3], dtype=dtype), T1=tensor(1.2, dtype=dtype),
                          T2=tensor(0.05, dtype=dtype), Δf=df[:1]),
        'norelax':   dict(M=M, dur=tensor(3e-3, dtype=dtype), T1=None, T2=None, Δf=df),
        'noprec':    dict(M=M, dur=tensor(3e-3, dtype=dtype), T1=t1, T2=t2, Δf=None),
    }


def mask_case(dtype, Nd=(5, 6, 7), N: int = 2):
    r"""a closed-form mask on an ``Nd`` grid (about 3/4 of the voxels), spatial
    tensors of the shapes mobjs moves through extract/embed (M: xyz, a map: scalar, b1Map:
    (xy, nCoils)), and FOV/offset per batch entry."""
    ix, iy, iz = torch.meshgrid(*[torch.arange(n) for n in Nd], indexing='ij')
    mask = (((ix * 7 + iy * 3 + iz * 5) % 4) != 0)[None]
    nV = Nd[0] * Nd[1] * Nd[2]
    nM = int(mask.sum())
    f64 = torch.float64

    def wave(shape, a):
        # exactly representable values (multiples of 1/64 in [-8, 8)): identical bits on any host
        n = 1
        for d in shape:
            n *= d
        k = torch.arange(n, dtype=torch.int64)
        return (((k * a + 11) % 1021 - 510).to(f64) / 64).reshape(shape).to(dtype)
    spatial = {'M': wave((N,) + Nd + (3,), 37), 'map': wave((N,) + Nd, 61),
               'b1': wave((N,) + Nd + (2, 4), 13)}
    compact = {'M': wave((N, nM, 3), 41), 'map': wave((N, nM), 29),
               'b1': wave((N, nM, 2, 4), 17)}
    fov = torch.tensor([[24., 20., 7.5], [3., 30., 12.]], dtype=f64)[:N].to(dtype)
    ofst = torch.tensor([[0., 0.5, -1.25], [2., 0., 0.125]], dtype=f64)[:N].to(dtype)
    return dict(shape=(N,) + Nd, Nd=Nd, N=N, nV=nV, nM=nM, mask=mask, spatial=spatial,
                compact=compact, fov=fov, ofst=ofst)
