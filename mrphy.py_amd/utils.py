r"""The one helper of ``mrphy.utils`` that sits on the hot path: Rodrigues rotation
(reference ``mrphy/utils.py:333-359``, used by ``slowsims.blochsim_1step``)."""
from math import prod

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host

__all__ = ['uϕrot']


class _UPhiRot(Function):
    r"""``Vo = _UPhiRot.apply(U, Φ, Vi)`` with the explicit adjoint ``mrphy_uphirot_bwd`` (the
    reference differentiates the plain torch expression, ``utils.py:351-357``)."""

    @staticmethod
    def forward(ctx, U, Φ, Vi):
        lib = _lib.require_library()
        dtype, device = Vi.dtype, Vi.device
        has_nv = Vi.ndim == U.ndim + 1
        rows = prod(U.shape[:-1])
        nV = Vi.shape[-1] if has_nv else 1
        Uc = U.detach().to(dtype).contiguous()
        Φc = Φ.detach().to(device=device, dtype=dtype).expand(U.shape[:-1]).contiguous()
        Vc = Vi.detach().contiguous()
        Vo = torch.empty_like(Vc)
        code = _lib.F64 if dtype == torch.float64 else _lib.F32
        with torch.cuda.device(device):
            rc = lib.mrphy_uphirot(code, Uc.data_ptr(), Φc.data_ptr(), Vc.data_ptr(), Vo.data_ptr(),
                                   rows, nV, _host.current_stream(device))
        _lib.check(rc, 'mrphy_uphirot')
        ctx.save_for_backward(Uc, Φc, Vc)
        ctx.meta = (code, rows, nV, U.shape, U.dtype, Φ.shape, Φ.dtype)
        return Vo

    @staticmethod
    def backward(ctx, gVo):
        lib = _lib.require_library()
        Uc, Φc, Vc = ctx.saved_tensors
        code, rows, nV, U_shape, U_dtype, Φ_shape, Φ_dtype = ctx.meta
        need_U, need_Φ, need_V = ctx.needs_input_grad
        g = gVo.to(Vc.dtype).contiguous()
        gU = torch.empty_like(Uc) if need_U else None
        gΦ = torch.empty_like(Φc) if need_Φ else None
        gV = torch.empty_like(Vc) if need_V else None
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(Vc.device):
            rc = lib.mrphy_uphirot_bwd(code, Uc.data_ptr(), Φc.data_ptr(), Vc.data_ptr(), g.data_ptr(),
                                       ptr(gU), ptr(gΦ), ptr(gV), rows, nV,
                                       _host.current_stream(Vc.device))
        _lib.check(rc, 'mrphy_uphirot_bwd')
        if need_U:
            gU = gU.to(U_dtype).reshape(U_shape)
        if need_Φ:                      # Φ may have been broadcast over (N, *Nd)
            from .beffective import _sum_to
            gΦ = (_sum_to(gΦ, Φ_shape) if tuple(Φ_shape) != tuple(gΦ.shape) else gΦ).to(Φ_dtype)
        return gU, gΦ, gV


def uϕrot(U: Tensor, Φ: Tensor, Vi: Tensor) -> Tensor:
    r"""Rotate ``Vi`` about axis ``U`` by ``Φ`` (``utils.py:333-359``).

    ``Vo = cosΦ·Vi + (1-cosΦ)·(U·Vi)·U + sinΦ·U×Vi``

    Inputs:
        - ``U``:  `(N, *Nd, xyz)`, rotation axes, assumed unit;
        - ``Φ``:  `(N, *Nd,)`, rotation angles;
        - ``Vi``: `(N, *Nd, xyz, (nV))`, vectors to be rotated.
    Outputs:
        - ``Vo``: `(N, *Nd, xyz, (nV))`.

    Differentiable w.r.t. ``U``, ``Φ`` and ``Vi`` (explicit adjoint kernel), like the reference's
    plain torch expression.
    """
    _host.require_device_tensor(U, 'U')
    _host.require_device_tensor(Vi, 'Vi')
    has_nv = Vi.ndim == U.ndim + 1
    assert has_nv or Vi.shape == U.shape
    return _UPhiRot.apply(U, Φ, Vi)
