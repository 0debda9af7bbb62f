r"""Fused ``rf, gr -> Mo``: ``rfgr2beff`` + ``blochsim`` in one kernel, no ``Beff`` in HBM.

This is what ``mrphy.mobjs.SpinArray.applypulse`` does back-to-back (reference
``mobjs.py:435-446``: ``pulse2beff`` then ``sims.blochsim``).  The per-step field is assembled
in registers from the wave-uniform pulse sample and the lane's own ``loc, Δf/γ, b1Map``
(``beffective.py:137-165``), then the same step as ``mrphy_blochsim_fwd`` is applied.

Gradients w.r.t. ``rf``/``gr``/``Mi`` currently go through the materialised kernels
(K0 + K1 + K3 + K0-adjoint, all HIP): when any of them requires grad, :func:`blochsim_rfgr`
composes ``rfgr2beff`` and ``blochsim`` eagerly.
"""
from math import pi as π, prod  # noqa: F401
from typing import Optional

import torch
from torch import Tensor

from . import _lib, _host
from ._consts import γH, dt0

__all__ = ['blochsim_rfgr']


def blochsim_rfgr(
    Mi: Tensor, rf: Tensor, gr: Tensor, loc: Tensor, *,
    Δf: Optional[Tensor] = None, b1Map: Optional[Tensor] = None, γ_beff: Tensor = γH,
    T1: Optional[Tensor] = None, T2: Optional[Tensor] = None,
    γ: Tensor = γH, dt: Tensor = dt0, consts: Optional[dict] = None
) -> Tensor:
    r"""``blochsim(Mi, rfgr2beff(rf, gr, loc, Δf=Δf, b1Map=b1Map, γ=γ_beff), T1=T1, T2=T2,
    γ=γ, dt=dt)`` without the intermediate tensor.

    ``Mi``: `(N, *Nd, xyz)`; the other arguments as in
    :func:`mrphy_amd.beffective.rfgr2beff` and :func:`mrphy_amd.sims.blochsim`.  ``consts``
    (``dict(γ2πdt=, E1=, E1_1=, E2=)``) replaces ``T1, T2, γ, dt`` as in
    :func:`mrphy_amd.sims.blochsim_consts`.
    """
    from . import beffective, sims
    _host.require_device_tensor(Mi, 'Mi')
    assert (T1 is None) == (T2 is None)
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(x, Tensor) and x.requires_grad for x in (Mi, rf, gr, loc, Δf, b1Map))
    if needs_grad:
        beff = beffective.rfgr2beff(rf, gr, loc, Δf=Δf, b1Map=b1Map, γ=γ_beff, lazy=False)
        if consts is not None:
            return sims.blochsim_consts(Mi, beff, **consts)
        return sims.blochsim(Mi, beff, T1=T1, T2=T2, γ=γ, dt=dt)

    lib = _lib.require_library()
    p = beffective._PulseOnSpins(rf.detach(), gr.detach(), loc.detach(),
                                 None if Δf is None else Δf.detach(),
                                 None if b1Map is None else b1Map.detach(), γ_beff.detach())
    device, dtype = Mi.device, Mi.dtype
    assert p.device == device and p.dtype == dtype, "Mi and loc must share device and dtype"
    assert tuple(Mi.shape[:-1]) == (p.N,) + p.Nd
    ndim = 1 + len(p.Nd) + 2
    cdev = _host.const_device(device)
    mv = lambda x: None if x is None else _host.pad_trailing(x.to(cdev), ndim)  # noqa: E731
    if consts is not None:
        γ2πdt, E1, E2, E1_1 = (consts.get(k) for k in ('γ2πdt', 'E1', 'E2', 'E1_1'))
    else:
        γ2πdt, E1, E2, E1_1 = sims._gamma_dt_constants(mv(T1), mv(T2), mv(γ), mv(dt))
    code, g, e1, e2, e1m1 = sims._prep_constants(γ2πdt, E1, E2, E1_1, p.N, p.Nd, dtype, device)
    Mi_c = Mi.detach().contiguous()
    Mo = torch.empty_like(Mi_c)
    nul = _host.NULL_BC
    with torch.cuda.device(device):
        rc = lib.mrphy_blochsim_rfgr_fwd(
            code, Mi_c.data_ptr(), *p.k0_args(), *g.args,
            *(e1.args if e1 else nul), *(e2.args if e2 else nul),
            e1m1.t.data_ptr() if e1m1 else None,
            Mo.data_ptr(), None, 0, p.N, p.nM, p.nT, p.nC, _host.current_stream(device))
    _lib.check(rc, 'mrphy_blochsim_rfgr_fwd')
    return Mo
