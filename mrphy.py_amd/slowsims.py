r"""Single-step Bloch simulation, HIP-backed.

Drop-in for ``mrphy.slowsims.blochsim_1step`` (reference ``mrphy/slowsims.py:15-54``).
``slowsims.blochsim`` (``slowsims.py:57-114``) is the reference's own parity oracle for
``sims.blochsim``; here both names run the same kernel.
"""
from math import prod
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host, sims
from ._consts import γH, dt0

__all__ = ['blochsim_1step', 'blochsim', 'blochsim_ab', 'freeprec']


@_host.half_via_float
def blochsim_1step(
    M: Tensor,
    M1: Tensor,
    b: Tensor,
    E1: Tensor,
    E1_1: Tensor,
    E2: Tensor,
    γ2πdt: Tensor,
) -> Tuple[Tensor, Tensor]:
    r"""Single step bloch simulation (``slowsims.py:15-54``).

    Usage:
        ``M, M_old = blochsim_1step(M, M1, b, E1, E1_1, E2, γ2πdt)``
    Inputs:
        - ``M``: `(N, *Nd, xyz)`, spins.
        - ``M1``: `(N, *Nd, xyz)`, ignored, as in the reference (it rebinds it at once).
        - ``b``: `(N, *Nd, xyz)`, "Gauss", B-effective of this step.
        - ``E1``, ``E1_1``, ``E2``, ``γ2πdt``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`.
    Outputs:
        - ``(M_new, M)``: the stepped spins and the input (the reference returns its two
          buffers swapped).  Differentiable w.r.t. ``M``, ``b`` and the four constants through the
          explicit adjoint kernel (the reference: autograd over ``beff2uϕ``/``uϕrot``,
          ``slowsims.py:42-54``).
    """
    _host.require_device_tensor(M, 'M')
    _host.require_device_tensor(b, 'b')
    assert (M.shape == b.shape)
    # one step of the integrator == sims.blochsim over Beff (N, *Nd, 1, xyz): the same kernel
    # (mrphy_blochsim_fwd with nT = 1 is what mrphy_blochsim_1step launches); when M, b or a constant
    # require grad the autograd pair of sims.blochsim supplies the explicit adjoint -- for the
    # constants too (round 3: mrphy_blochsim_bwd_consts), as autograd does in the reference
    if sims._wants_grad(M, b, E1, E1_1, E2, γ2πdt):
        Mn = sims.BlochSimHIP.apply(M, b.unsqueeze(-2), γ2πdt, E1, E2, E1_1, True)
        return Mn, M
    lib = _lib.require_library()
    device, dtype = M.device, M.dtype
    N, Nd = M.shape[0], tuple(M.shape[1:-1])
    code, g, e1, e2, e1m1 = sims._prep_constants(γ2πdt, E1, E2, E1_1, N, Nd, dtype, device)
    Mc, bc = M.detach().contiguous(), b.detach().to(dtype).contiguous()
    Mn = torch.empty_like(Mc)
    with torch.cuda.device(device):
        rc = lib.mrphy_blochsim_1step(code, Mc.data_ptr(), bc.data_ptr(), *g.args, *e1.args, *e2.args,
                                      e1m1.t.data_ptr(), Mn.data_ptr(), N, prod(Nd),
                                      _host.current_stream(device))
    _lib.check(rc, 'mrphy_blochsim_1step')
    return Mn, M


@_host.half_via_float
def blochsim(
    M: Tensor,
    Beff: Tensor, *,
    T1: Optional[Tensor] = None,
    T2: Optional[Tensor] = None,
    γ: Tensor = γH,
    dt: Tensor = dt0
) -> Tensor:
    r"""``mrphy.slowsims.blochsim`` (``slowsims.py:57-114``): same physics as
    :func:`mrphy_amd.sims.blochsim`, which it forwards to -- except that, like the reference's
    (plain differentiable torch ops, ``slowsims.py:86-112``), it is differentiable w.r.t. ``T1, T2, γ,
    dt`` as well: when one of them requires grad the constants are formed with differentiable torch
    ops (the reference's expressions) and the adjoint sweep returns their gradients too
    (``mrphy_blochsim_bwd_consts``)."""
    if sims._wants_grad(T1, T2, γ, dt):
        from .beffective import LazyBeff
        assert (M.shape[:-1] == Beff.shape[:-2])
        assert ((T1 is None) == (T2 is None))
        _host.require_device_tensor(M, 'M')
        if isinstance(Beff, LazyBeff):
            Beff = Beff.materialize()
        Beff = Beff.to(M.device)
        pad = lambda x: None if x is None else _host.pad_trailing(x.to(M.device), Beff.ndim)  # noqa: E731
        γ2πdt, E1, E2, E1_1 = sims._gamma_dt_constants(pad(T1), pad(T2), pad(γ), pad(dt))
        return sims.BlochSimHIP.apply(M, Beff, γ2πdt, E1, E2, E1_1, True)
    return sims.blochsim(M, Beff, T1=T1, T2=T2, γ=γ, dt=dt)


@_host.half_via_float
def freeprec(
    M: Tensor, dur: Tensor, *,
    T1: Optional[Tensor] = None, T2: Optional[Tensor] = None,
    Δf: Optional[Tensor] = None
) -> Tensor:
    r"""``mrphy.slowsims.freeprec`` (``slowsims.py:134-174``): same physics as
    :func:`mrphy_amd.sims.freeprec` and the same kernel -- and, like the reference's plain torch ops
    (``slowsims.py:151-174``), differentiable w.r.t. ``dur``, ``T1``, ``T2``, ``Δf`` as well (round 4:
    ``mrphy_freeprec_bwd_consts``, per spin, summed over the axes each operand broadcasts along)."""
    assert ((T1 is None) == (T2 is None))  # both or neither
    _host.require_device_tensor(M, 'M')
    if sims._wants_grad(dur, T1, T2, Δf):
        return sims.FreePrecHIP.apply(M, dur, T1, T2, Δf, True)
    return sims.freeprec(M, dur, T1=T1, T2=T2, Δf=Δf)


class _BlochSimAB(Function):
    @staticmethod
    def forward(ctx, M, A, B):
        lib = _lib.require_library()
        dtype, device = M.dtype, M.device
        Mc, Ac, Bc_ = (x.detach().to(dtype).contiguous() for x in (M, A, B))
        Mo = torch.empty_like(Mc)
        rows = Mc.numel() // 3
        code = _lib.F64 if dtype == torch.float64 else _lib.F32
        with torch.cuda.device(device):
            rc = lib.mrphy_blochsim_ab(code, Mc.data_ptr(), Ac.data_ptr(), Bc_.data_ptr(),
                                       Mo.data_ptr(), rows, _host.current_stream(device))
        _lib.check(rc, 'mrphy_blochsim_ab')
        ctx.save_for_backward(Mc, Ac)
        ctx.code = code
        return Mo

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_library()
        Mc, Ac = ctx.saved_tensors
        need_M, need_A, need_B = ctx.needs_input_grad
        gc = g.to(Mc.dtype).contiguous()
        gM = torch.empty_like(Mc) if need_M else None
        gA = torch.empty_like(Ac) if need_A else None
        if need_M or need_A:
            with torch.cuda.device(Mc.device):
                rc = lib.mrphy_blochsim_ab_bwd(ctx.code, Mc.data_ptr(), Ac.data_ptr(), gc.data_ptr(),
                                               gM.data_ptr() if need_M else None,
                                               gA.data_ptr() if need_A else None,
                                               Mc.numel() // 3, _host.current_stream(Mc.device))
            _lib.check(rc, 'mrphy_blochsim_ab_bwd')
        return gM, gA, (gc if need_B else None)


@_host.half_via_float
def blochsim_ab(M: Tensor, A: Tensor, B: Tensor) -> Tensor:
    r"""Bloch simulation via Hargreaves' mat/vec representation (``slowsims.py:117-131``).

    Usage:
        ``M = blochsim_ab(M, A, B)``
    Inputs:
        - ``M``: `(N, *Nd, xyz)`, spins, assumed equilibrium magnitude [0 0 1];
        - ``A``: `(N, *Nd, xyz, 3)`; ``B``: `(N, *Nd, xyz)`
          (:func:`mrphy_amd.beffective.beff2ab`).
    Outputs:
        - ``M``: `(N, *Nd, xyz)`, ``A @ M + B``.
    """
    _host.require_device_tensor(M, 'M')
    _host.require_device_tensor(A, 'A')
    assert A.shape == M.shape + (3,) and B.shape == M.shape
    return _BlochSimAB.apply(M, A, B)
