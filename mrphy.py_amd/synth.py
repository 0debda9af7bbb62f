r"""Synthetic workloads: the closed-form inputs every parity test, golden fixture and bench
line uses.  Everything is a formula of the spin index and the time index, so
that the build container (which runs the reference on a subset) and the GPU box (which runs
the full grid) produce the same numbers without shipping data.

Geometry is ``mrphy.mobjs.SpinCube((1,n,n,n), fov=[[24,24,24]], ofst=0)`` with an all-true mask:
``loc[s] = fov*(i - n//2)/n`` per axis (reference ``mobjs.py:828-837``, ``utils.py:27-33``),
compact index ``s = (ix*n + iy)*n + iz``.
"""
from math import pi as π

import torch

from ._consts import γH

FOV = 24.0   # cm


def cube_index(n: int, idx: torch.Tensor = None, device='cpu'):
    r"""(ix, iy, iz) of compact spin indices ``idx`` (default: all ``n**3``) of an n-cube."""
    if idx is None:
        idx = torch.arange(n ** 3, device=device)
    idx = idx.to(torch.int64)
    return idx // (n * n), (idx // n) % n, idx % n


def cube_spins(n: int, idx: torch.Tensor = None, *, dtype=torch.float32, device='cpu',
               seed_M0: int = None):
    r"""Per-spin maps of the synthetic cube, each shaped ``(1, nM[, ...])``.

    Returns a dict: ``loc`` (1,nM,3) cm, ``T1``/``T2`` (1,nM) s, ``Δf`` (1,nM) Hz,
    ``γ`` (1,1) Hz/G, ``M0`` (1,nM,3).  Maps are evaluated in fp64 and rounded once to
    ``dtype``.  ``M0`` is ``[0,0,1]`` unless ``seed_M0`` is given: then uniform ``[0,1)^3``
    drawn on the CPU from that seed for the FULL cube and indexed by ``idx`` (so a subset sees
    the same values the full grid does).
    """
    ix, iy, iz = cube_index(n, idx, device)
    f64 = torch.float64
    ax = lambda i: FOV * (i.to(f64) - (n // 2)) / n  # noqa: E731
    x, y, z = ax(ix), ax(iy), ax(iz)
    loc = torch.stack([x, y, z], dim=-1)[None]
    T1 = (1.0 + 0.4 * torch.sin(2 * π * x / FOV))[None]
    T2 = (0.06 + 0.03 * torch.cos(2 * π * y / FOV))[None]
    df = (100.0 * torch.sin(2 * π * x / FOV) * torch.cos(2 * π * z / FOV))[None]
    nM = x.numel()
    if seed_M0 is None:
        M0 = torch.zeros((1, nM, 3), dtype=f64, device=x.device)
        M0[..., 2] = 1.0
    else:
        gen = torch.Generator(device='cpu').manual_seed(seed_M0)
        full = torch.rand((n ** 3, 3), generator=gen, dtype=f64)
        sel = full if idx is None else full[idx.cpu().to(torch.int64)]
        M0 = sel[None].to(x.device)
    cast = lambda t: t.to(dtype=dtype, device=device)  # noqa: E731
    return {'loc': cast(loc), 'T1': cast(T1), 'T2': cast(T2), 'Δf': cast(df),
            'γ': cast(γH.reshape(1, 1)), 'M0': cast(M0)}


def pulse(nT: int, *, dtype=torch.float32, device='cpu', dt: float = 4e-6):
    r"""Synthetic pulse (generalises ``mobjs.Examples.pulse``, ``mobjs.py:987-996``, to
    amplitudes within the hardware limits ``rfmax0``, ``gmax0``):

    ``rf = 0.2·[cos(2πt/nT), sin(2πt/nT)]`` G `(1,2,nT)`,
    ``gr = [0.5, 0.5, (2/π)·atan(t - nT/2)]`` G/cm `(1,3,nT)``, ``dt`` `(1,)` s.
    """
    t = torch.arange(nT, dtype=torch.float64, device=device)
    rf = 0.2 * torch.stack([torch.cos(2 * π * t / nT), torch.sin(2 * π * t / nT)])[None]
    gr = torch.stack([torch.full_like(t, 0.5), torch.full_like(t, 0.5),
                      (2 / π) * torch.atan(t - nT / 2)])[None]
    return {'rf': rf.to(dtype), 'gr': gr.to(dtype),
            'dt': torch.tensor([dt], dtype=dtype, device=device)}


def subset_indices(n: int, count: int, seed: int) -> torch.Tensor:
    r"""``count`` distinct compact spin indices of the n-cube, sorted, from ``seed`` (CPU RNG:
    identical on every machine)."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randperm(n ** 3, generator=gen)[:count].sort().values


# BASELINE.json configs (index = position in `configs`)
CONFIGS = {
    1: dict(n=64, nT=1024),     # 64^3 x 1024, 1 GPU
    2: dict(n=128, nT=4096),    # 128^3 x 4096, 1 GPU: the headline
    3: dict(n=128, nT=4096),    # same, sharded over 8 GPUs
    4: dict(n=64, nT=2048),     # 64^3 x 2048 after interpT (coarse 1024 @ 8e-6 -> 4e-6), fwd+bwd
}
