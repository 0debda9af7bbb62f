r"""CPU oracle for the Bloch-simulation hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file, and only as the checker / the timed CPU baseline.  Nothing under ``mrphy.py_amd/``
imports it; the product path has no CPU fallback.

It restates, in plain PyTorch CPU ops, the algorithm of the reference tianrluo/MRphy.py v0.2.0
for the one path this repository accelerates.  Each function cites the reference lines it
follows.  Two forms of the simulator are kept on purpose, as the reference keeps two:

* :func:`blochsim_slow` -- out-of-place ops, gradients by autograd (the reference's
  ``slowsims.blochsim``, which the reference itself uses as the oracle of ``sims``);
* :func:`blochsim` -- the explicit forward + hand-derived adjoint of ``sims.BlochSim``.  Its
  forward issues the SAME sequence of ATen calls per time step on the same strided history
  views as the reference, because ``bench.py`` times it as "the reference's
  CPU PyTorch path".

PINNING: ``tests/golden/make_golden.py --check`` (run in the build container, where the
reference is importable) compares every function here with the imported reference;
``tests/test_oracle_golden.py`` compares them with the committed golden vectors
(``tests/golden/*.npz``: outputs of the reference itself plus the known answers hard-coded in
the reference's tests, ``tests/test_slowsims.py:77-80``, ``tests/test_mobjs.py:112-120``).
"""
from math import pi as π
from typing import Optional, Tuple

import torch
from torch import Tensor, tensor
from torch.autograd import Function

# reference mrphy/__init__.py:58-62 (0-dim doubles)
γH = tensor(4257.6, dtype=torch.double)
dt0 = tensor(4e-6, dtype=torch.double)


def _rpad(x: Tensor, rank: int) -> Tensor:
    """Right-pad the shape with ones up to ``rank`` (the reference's reshape idiom)."""
    return x.reshape(tuple(x.shape) + (rank - x.ndim) * (1,))


# =============================================================================================
# rfgr2beff  -- reference mrphy/beffective.py:107-168
# =============================================================================================
def rfgr2beff(rf: Tensor, gr: Tensor, loc: Tensor, *, Δf: Optional[Tensor] = None,
              b1Map: Optional[Tensor] = None, γ: Tensor = γH) -> Tensor:
    r"""``rf (N,xy,nT[,nC])``, ``gr (N,xyz,nT)``, ``loc (N,*Nd,xyz)`` -> ``beff (N,*Nd,nT,xyz)``."""
    assert rf.device == gr.device == loc.device                       # :131
    N, Nd = loc.shape[0], tuple(loc.shape[1:-1])
    k = len(Nd)

    # z: gradient field at each location, a batched (nM,3)@(3,nT) product (:137)
    Bz = torch.matmul(loc.reshape(N, -1, 3), gr).reshape((N,) + Nd + (-1,))
    if Δf is not None:                                                # :139-142
        Bz += _rpad(Δf, k + 2) / _rpad(γ.to(loc.device), k + 2)

    rf = rf.reshape((-1,) + k * (1,) + tuple(rf.shape[1:]))           # :145
    if b1Map is None:                                                 # :147-151
        if rf.ndim == Bz.ndim + 2:
            rf = rf.sum(dim=-1)
        Bx, By = rf[..., 0, :].expand_as(Bz), rf[..., 1, :].expand_as(Bz)
    else:                                                             # :153-165
        if b1Map.ndim == k + 2:
            b1Map = b1Map[..., None]
        if rf.ndim == b1Map.ndim:
            rf = rf[..., None]
        b1 = b1Map.to(loc.device)[..., None, :]                       # (N,*Nd,xy,1,nC)
        b1_re, b1_im = b1[..., 0, :, :], b1[..., 1, :, :]
        rf_re, rf_im = rf[..., 0, :, :], rf[..., 1, :, :]
        # complex product b1*rf summed over coils
        Bx = (b1_re * rf_re - b1_im * rf_im).sum(dim=-1).expand_as(Bz)
        By = (b1_re * rf_im + b1_im * rf_re).sum(dim=-1).expand_as(Bz)
    return torch.stack((Bx, By, Bz), dim=-1)                          # :167


# =============================================================================================
# beff2uϕ -- reference mrphy/beffective.py:18-37;  uϕrot -- reference mrphy/utils.py:333-359
# =============================================================================================
def beff2uphi(beff: Tensor, γ2πdt: Tensor, *, dim: int = -1) -> Tuple[Tensor, Tensor]:
    nrm = torch.norm(beff, dim=dim)
    U = beff / nrm.clamp(min=1e-12).unsqueeze(dim)    # F.normalize(eps=1e-12) (:35)
    return U, -nrm * γ2πdt                            # sign: M x B (:36)


def uphirot(U: Tensor, Φ: Tensor, Vi: Tensor) -> Tensor:
    if Vi.ndim == U.ndim:
        ax, Φ, U = -1, Φ[..., None], U
    else:                                             # Vi (..., xyz, nV) (:351-352)
        ax, Φ, U = -2, Φ[..., None, None], U[..., None]
    c, s = torch.cos(Φ), torch.sin(Φ)
    along = (U * Vi).sum(dim=ax, keepdim=True)
    return c * Vi + (1 - c) * along * U + s * torch.cross(U.expand_as(Vi), Vi, dim=ax)


# =============================================================================================
# slowsims -- reference mrphy/slowsims.py:15-54 (1 step), :57-114 (all steps)
# =============================================================================================
def _relax_(M1: Tensor, E1: Tensor, E1_1: Tensor, E2: Tensor):
    """In-place relaxation of the rotated spins (slowsims.py:49-51 / 108-110)."""
    M1[..., 0:2] *= E2
    M1[..., 2] *= E1
    M1[..., 2] -= E1_1


def blochsim_1step(M: Tensor, M1: Tensor, b: Tensor, E1: Tensor, E1_1: Tensor, E2: Tensor,
                   γ2πdt: Tensor) -> Tuple[Tensor, Tensor]:
    r"""One step; returns ``(M_new, M_old)``.  As in the reference, when every rotation angle
    is zero the input tensor itself is relaxed in place and returned (slowsims.py:44-47)."""
    u, ϕ = beff2uphi(b, γ2πdt)
    Mn = uphirot(u, ϕ, M) if torch.any(ϕ != 0) else M
    _relax_(Mn, E1, E1_1, E2[..., None])
    return Mn, M


def blochsim_slow(M: Tensor, Beff: Tensor, *, T1: Optional[Tensor] = None,
                  T2: Optional[Tensor] = None, γ: Tensor = γH, dt: Tensor = dt0) -> Tensor:
    r"""Out-of-place simulator; autograd supplies the Jacobian (slowsims.py:57-114)."""
    assert M.shape[:-1] == Beff.shape[:-2]
    dev, k = M.device, M.ndim - 1
    one = tensor(1, device=dev, dtype=M.dtype)
    E1 = one if T1 is None else torch.exp(-dt / T1.to(dev))          # :90-91
    E2 = one if T2 is None else torch.exp(-dt / T2.to(dev))
    Beff, γ, dt = Beff.to(dev), γ.to(dev), dt.to(dev)
    E1, E2, γ, dt = (_rpad(x, k) for x in (E1, E2, γ, dt))            # :95-96
    E1_1, E2, g = E1 - 1, E2[..., None], 2 * π * γ * dt                # :98
    for t in range(Beff.shape[-2]):
        u, ϕ = beff2uphi(Beff[..., t, :], g)
        Mn = uphirot(u, ϕ, M) if torch.any(ϕ != 0) else M
        _relax_(Mn, E1, E1_1, E2)
        M = Mn
    return M


# =============================================================================================
# sims.BlochSim -- reference mrphy/sims.py:32-132 (forward), :135-269 (backward)
# =============================================================================================
class _History:
    """What the explicit forward leaves for its adjoint (sims.py:84-88,128-130)."""
    __slots__ = ('Mi', 'M', 'U', 'Phi', 'C1', 'S', 'UM', 'E', 'e1m1', 'g')


def explicit_forward(Mi: Tensor, Beff: Tensor, T1, T2, γ, dt) -> Tuple[Tensor, _History]:
    r"""Forward with history.  Per time step this issues, in order and on ``(…,1,k)`` strided
    views of ``(…,nT,k)`` history tensors: norm, clamp_, div, sin, cos, sub_, mul, sum, cross,
    addcmul x3, (mul_, sub_) -- the 14 ATen launches of sims.py:100-124."""
    kw = dict(dtype=Mi.dtype, device=Mi.device)
    lead, nT = tuple(Beff.shape[:-2]), Beff.shape[-2]
    h = _History()
    h.g = 2 * π * γ * dt                                               # :62
    h.U = torch.empty(Beff.shape, **kw)
    torch.mul(h.g, Beff, out=h.U)                                      # :63-64 (γBeff, later u)
    assert (T1 is None) == (T2 is None)                                # :68
    if T1 is None:
        h.E = h.e1m1 = None
    else:                                                              # :74-76
        e1, e2 = -dt / T1, -dt / T2
        e1.exp_(), e2.exp_()
        h.E, h.e1m1 = torch.cat((e2, e2, e1), dim=-1), e1 - 1
    h.Mi = Mi.clone(memory_format=torch.contiguous_format)[..., None, :]
    h.M = torch.empty(lead + (nT, 3), **kw)
    h.Phi, h.C1, h.S, h.UM = (torch.empty(lead + (nT, 1), **kw) for _ in range(4))
    scratch = torch.empty(h.Mi.shape, **kw)

    prev = h.Mi
    for t in range(nT):
        cur, u = h.M.narrow(-2, t, 1), h.U.narrow(-2, t, 1)
        ϕ, c1 = h.Phi.narrow(-2, t, 1), h.C1.narrow(-2, t, 1)
        s, um = h.S.narrow(-2, t, 1), h.UM.narrow(-2, t, 1)
        torch.norm(u, dim=-1, keepdim=True, out=ϕ)                     # :100
        ϕ.clamp_(min=1e-12)                                            # :101
        torch.div(u, ϕ, out=u)                                         # :102 (axis overwrites γB)
        torch.sin(ϕ, out=s)                                            # :105
        torch.cos(ϕ, out=c1)                                           # :106
        c1.sub_(1)                                                     # :107
        torch.mul(u, prev, out=cur)                                    # :113
        torch.sum(cur, dim=-1, keepdim=True, out=um)                   # :114
        torch.cross(u, prev, dim=-1, out=cur)                          # :116
        torch.addcmul(prev, s, cur, value=-1, out=cur)                 # :117
        torch.addcmul(prev, um, u, value=-1, out=scratch)              # :119
        torch.addcmul(cur, c1, scratch, out=cur)                       # :121
        if h.E is not None:                                            # :77,124
            cur.mul_(h.E)[..., 2:3].sub_(h.e1m1)
        prev = cur
    return h.M[..., -1, :].clone(), h                                  # :131


def explicit_backward(h: _History, grad_Mo: Tensor, need_Mi: bool = True, need_B: bool = True):
    r"""Adjoint sweep (sims.py:195-261), written out of place:

    ``h̃ = E⊙h``;  ``m̃₁ = (m₁ + (0,0,E1-1))/E``;
    ``h₀ = h̃ + (cϕ-1)(h̃ - (u·h̃)u) + sϕ u×h̃``                                   (:216-227)
    ``∂L/∂B = -γ2πdt·{ sϕ/ϕ (m₀×h̃) + (cϕ-1)/ϕ ((u·m₀)h̃ + (u·h̃)m₀)
                       - [(m̃₁ - sϕ/ϕ m₀)·(u×h̃) + 2(cϕ-1)/ϕ (u·h̃)(u·m₀)] u }``       (:229-259)
    (the reference folds ``-γ2πdt`` into ``h`` up front, :194, and divides it back out of
    ``grad_Mi`` with ``γ2πdt[0, ...]``, :267 -- which is wrong for per-spin γ; here the factor
    is applied where it belongs).
    """
    nT = h.M.shape[-2]
    hv = grad_Mo.clone()[..., None, :]
    gB = torch.empty_like(h.U) if need_B else None
    zero_e = None
    if h.E is not None:
        zero_e = torch.cat((torch.zeros_like(h.e1m1), torch.zeros_like(h.e1m1), h.e1m1), dim=-1)
    for t in range(nT - 1, -1, -1):
        m0 = h.Mi if t == 0 else h.M.narrow(-2, t - 1, 1)
        m1 = h.M.narrow(-2, t, 1)
        u, ϕ = h.U.narrow(-2, t, 1), h.Phi.narrow(-2, t, 1)
        c1, s, um = h.C1.narrow(-2, t, 1), h.S.narrow(-2, t, 1), h.UM.narrow(-2, t, 1)
        if h.E is not None:
            ht, mt = hv * h.E, (m1 + zero_e) / h.E                     # :172-177
        else:
            ht, mt = hv, m1
        uh = (u * ht).sum(dim=-1, keepdim=True)
        uxh = torch.cross(u, ht, dim=-1)
        if need_B:
            sp, cp = s / ϕ, c1 / ϕ                                     # :234
            along = ((mt - sp * m0) * uxh).sum(dim=-1, keepdim=True) + 2 * cp * uh * um
            d = sp * torch.cross(m0, ht, dim=-1) + cp * (um * ht + uh * m0) - along * u
            gB.narrow(-2, t, 1).copy_(-h.g * d)
        hv = ht + c1 * (ht - uh * u) + s * uxh
    return (hv[..., 0, :] if need_Mi else None), gB


class _ExplicitBloch(Function):
    @staticmethod
    def forward(ctx, Mi, Beff, T1, T2, γ, dt):
        Mo, hist = explicit_forward(Mi, Beff, T1, T2, γ, dt)
        ctx.hist = hist
        return Mo

    @staticmethod
    def backward(ctx, grad_Mo):
        need = ctx.needs_input_grad
        if not any(need[0:2]):                                         # :156-157
            return (None,) * 6
        gMi, gB = explicit_backward(ctx.hist, grad_Mo, need[0], need[1])
        return gMi, gB, None, None, None, None


def blochsim(Mi: Tensor, Beff: Tensor, *, T1: Optional[Tensor] = None,
             T2: Optional[Tensor] = None, γ: Tensor = γH, dt: Tensor = dt0) -> Tensor:
    r"""``sims.blochsim`` (sims.py:272-315): shape checks, right-padding of γ, dt, T1, T2 to
    the rank of ``Beff``, then the explicit forward/adjoint pair."""
    assert Mi.shape[:-1] == Beff.shape[:-2]                            # :305
    Beff, rank = Beff.to(Mi.device), Beff.ndim
    γ, dt = _rpad(γ, rank), _rpad(dt, rank)
    assert (T1 is None) == (T2 is None)                                # :311
    if T1 is not None:
        T1, T2 = _rpad(T1, rank), _rpad(T2, rank)
    return _ExplicitBloch.apply(Mi, Beff, T1, T2, γ, dt)


# =============================================================================================
# freeprec -- reference mrphy/sims.py:318-458 (explicit), mrphy/slowsims.py:134-174 (autograd)
# =============================================================================================
def freeprec_slow(M: Tensor, dur: Tensor, *, T1: Optional[Tensor] = None,
                  T2: Optional[Tensor] = None, Δf: Optional[Tensor] = None) -> Tensor:
    r"""Out-of-place form (slowsims.py:134-174): rotate about z by -2πΔf·dur, then relax."""
    rank = M.ndim
    dur = _rpad(dur, rank)
    Mx, My, Mz = M.split(1, dim=-1)
    if Δf is not None:
        ϕ = -(2 * π) * _rpad(Δf, rank) * dur                          # :161-162
        c, s = torch.cos(ϕ), torch.sin(ϕ)
        Mx, My = c * Mx - s * My, s * Mx + c * My
    assert (T1 is None) == (T2 is None)
    if T1 is not None:
        E1, E2 = torch.exp(-dur / _rpad(T1, rank)), torch.exp(-dur / _rpad(T2, rank))
        Mx, My, Mz = E2 * Mx, E2 * My, E1 * Mz + 1 - E1                # :171
    return torch.cat((Mx, My, Mz), dim=-1)


class _ExplicitFreePrec(Function):
    """sims.FreePrec (sims.py:325-421): in-place forward on a clone, hand-written adjoint."""

    @staticmethod
    def forward(ctx, Mi, dur, T1, T2, Δf):
        Mo = Mi.clone(memory_format=torch.contiguous_format)
        c = s = keep = None
        if Δf is not None:                                            # :348-360
            s = -(2 * π) * Δf * dur[..., 0]
            c = torch.cos(s)
            s.sin_()
            keep = Mo[..., 0].clone()
            Mo[..., 0].mul_(c)
            torch.addcmul(Mo[..., 0], s, Mo[..., 1], value=-1, out=Mo[..., 0])
            Mo[..., 1].mul_(c)
            torch.addcmul(Mo[..., 1], s, keep, out=Mo[..., 1])
        E1 = E2 = None
        assert (T1 is None) == (T2 is None)
        if T1 is not None:                                            # :366-371
            E1, E2 = -dur / T1, -dur / T2
            E1_1 = torch.expm1(E1)
            E1.exp_(), E2.exp_()
            Mo[..., 0:2].mul_(E2)
            Mo[..., 2:3].mul_(E1).sub_(E1_1)
        ctx.save_for_backward(c, s, E1, E2)
        return Mo

    @staticmethod
    def backward(ctx, grad_Mo):
        if not ctx.needs_input_grad[0]:
            return (None,) * 5
        c, s, E1, E2 = ctx.saved_tensors
        g = grad_Mo.clone(memory_format=torch.contiguous_format)
        if E1 is not None:                                            # :406-408
            g[..., 0:2].mul_(E2)
            g[..., 2:3].mul_(E1)
        if c is not None:                                             # :411-419
            gy = g[..., 1].clone()
            g[..., 1] = c * gy - s * g[..., 0]
            g[..., 0] = c * g[..., 0] + s * gy
        return g, None, None, None, None


def freeprec(Mi: Tensor, dur: Tensor, *, T1: Optional[Tensor] = None,
             T2: Optional[Tensor] = None, Δf: Optional[Tensor] = None) -> Tensor:
    r"""``sims.freeprec`` (sims.py:424-458): right-pad dur, T1, T2 to rank(M), Δf to rank(M)-1."""
    rank = Mi.ndim
    dur = _rpad(dur, rank)
    assert (T1 is None) == (T2 is None)
    if T1 is not None:
        T1, T2 = _rpad(T1, rank), _rpad(T2, rank)
    if Δf is not None:
        Δf = _rpad(Δf, rank - 1)
    return _ExplicitFreePrec.apply(Mi, dur, T1, T2, Δf)


# =============================================================================================
# Reference points that are not the reference's own arithmetic
# =============================================================================================
def blochsim_f64_arith(Mi: Tensor, Beff: Tensor, *, T1=None, T2=None, γ=γH, dt=dt0,
                       consts: Optional[dict] = None) -> Tensor:
    r"""The "exact arithmetic, same rounded constants" yardstick of SURVEY.md §8c: constants
    γ2πdt, E1, E2, E1-1 are formed in the INPUT dtype exactly as :func:`explicit_forward` forms
    them, then everything (inputs and constants) is widened to fp64 and integrated in fp64.
    For fp32 inputs this isolates arithmetic round-off from constant round-off.
    ``consts = dict(γ2πdt=, E1=, E1_1=, E2=)`` (rank of ``Beff``) supplies already-rounded
    constants instead, e.g. the ones stored with a golden vector."""
    rank = Beff.ndim
    f64 = torch.float64
    M = Mi.to(f64)
    if consts is not None:
        g = _rpad(consts['γ2πdt'], rank)
        relax = consts.get('E1') is not None
        if relax:
            e1, e2, e1m1 = (_rpad(consts[k], rank) for k in ('E1', 'E2', 'E1_1'))
    else:
        γ, dt = _rpad(γ.to(Mi.device), rank), _rpad(dt.to(Mi.device), rank)
        g = 2 * π * γ * dt
        relax = T1 is not None
        if relax:
            T1, T2 = _rpad(T1, rank), _rpad(T2, rank)
            e1, e2 = torch.exp(-dt / T1), torch.exp(-dt / T2)
            e1m1 = e1 - 1
    if relax:
        e1m1 = e1m1.to(f64)[..., 0, :]
        e1, e2 = e1.to(f64)[..., 0, :], e2.to(f64)[..., 0, :]
    gB = g.to(f64) * Beff.to(f64)
    for t in range(Beff.shape[-2]):
        b = gB[..., t, :]
        ϕ = b.norm(dim=-1, keepdim=True).clamp(min=1e-12)
        u = b / ϕ
        M = M - torch.sin(ϕ) * torch.cross(u, M, dim=-1) \
            + (torch.cos(ϕ) - 1) * (M - (u * M).sum(-1, keepdim=True) * u)
        if relax:
            M = torch.cat((M[..., 0:2] * e2, M[..., 2:3] * e1 - e1m1), dim=-1)
    return M


# ---------------------------------------------------------------------------------------------
# mask gather/scatter and cube locations (the steps either side of the path)
# ---------------------------------------------------------------------------------------------
def mask_extract(v: Tensor, mask: Tensor) -> Tensor:
    r"""``SpinArray.extract`` (mobjs.py:532-553): `(N, *Nd, ...)` -> `(N, nM, ...)`, the voxels
    where the `(1, *Nd)` mask is set, in row-major order."""
    N, nd = v.shape[0], mask.ndim - 1
    flat = v.reshape((N, -1) + tuple(v.shape[1 + nd:]))
    return flat[:, mask.reshape(-1)]


def mask_embed(v_: Tensor, mask: Tensor, out: Optional[Tensor] = None) -> Tensor:
    r"""``SpinArray.embed`` (mobjs.py:512-530): `(N, nM, ...)` -> `(N, *Nd, ...)`; a fresh output
    is NaN outside the mask, a given ``out`` keeps its values there."""
    N, Nd, tail = v_.shape[0], tuple(mask.shape[1:]), tuple(v_.shape[2:])
    res = (torch.full((N,) + Nd + tail, float('nan'), dtype=v_.dtype) if out is None else out)
    flat = res.reshape((N, -1) + tail)
    assert flat.data_ptr() == res.data_ptr()
    flat[:, mask.reshape(-1)] = v_
    return res


def cube_loc(mask: Tensor, fov: Tensor, ofst: Tensor) -> Tensor:
    r"""``SpinCube._update_loc_`` (mobjs.py:815-839): locations `(N, nM, xyz)` of the masked voxels
    of an `(nx, ny, nz)` grid, normalised coordinates ``(i - n//2)/n`` in ``[-0.5, 0.5)``."""
    dims, kw = tuple(mask.shape[1:]), dict(dtype=fov.dtype)
    axes = [(torch.arange(n, **kw) - (n // 2)) / n for n in dims]
    grids = torch.meshgrid(*axes, indexing='ij')
    sel = mask[0]
    cols = [fov[:, None, i] * grids[i][sel][None, :] + ofst[:, None, i] for i in range(3)]
    return torch.stack(cols, dim=-1)


# ---------------------------------------------------------------------------------------------
# Hargreaves' A/B propagation
# ---------------------------------------------------------------------------------------------
def beff2ab(beff: Tensor, *, E1: Tensor = tensor(0.), E2: Tensor = tensor(0.), γ: Tensor = γH,
            dt: Tensor = dt0) -> Tuple[Tensor, Tensor]:
    r"""``beffective.beff2ab`` (beffective.py:40-104): carry ``[I | 0]`` `(N, *Nd, xyz, 3+1)`
    through the nT steps -- rotate all four columns about ``u`` by ``ϕ``, scale rows x, y by E2
    and row z by E1, subtract ``E1 - 1`` from the z entry of the last column -- and split it."""
    k = beff.ndim - 2
    E1, E2, γ, dt = (_rpad(x.to(beff.device), k) for x in (E1, E2, γ, dt))     # :66-70
    g, E1_1 = 2 * π * γ * dt, E1 - 1                                             # :72-73
    E1c, E2c = E1[..., None], E2[..., None, None]
    NNd = tuple(beff.shape[:-2])
    AB = torch.zeros(NNd + (3, 4), dtype=beff.dtype, device=beff.device)
    AB[..., 0, 0] = AB[..., 1, 1] = AB[..., 2, 2] = 1                            # :79-82
    for t in range(beff.shape[-2]):
        u, ϕ = beff2uphi(beff[..., t, :], g)
        AB1 = uphirot(u, ϕ, AB) if torch.any(ϕ != 0) else AB                     # :88-91
        top, bot = AB1[..., 0:2, :] * E2c, AB1[..., 2, :] * E1c                  # :94-95
        last = bot[..., 3] - E1_1                                                # :96
        bot = torch.cat([bot[..., 0:3], last[..., None]], dim=-1)
        AB = torch.cat([top, bot[..., None, :]], dim=-2)
    return AB[..., 0:3], AB[..., 3]


def blochsim_ab(M: Tensor, A: Tensor, B: Tensor) -> Tensor:
    r"""``slowsims.blochsim_ab`` (slowsims.py:117-131): ``A @ M + B`` per spin."""
    return (A @ M[..., None]).squeeze(dim=-1) + B
