r"""The one helper of ``mrphy.utils`` that sits on the hot path: Rodrigues rotation
(reference ``mrphy/utils.py:333-359``, used by ``slowsims.blochsim_1step``)."""
from math import prod

import torch
from torch import Tensor

from . import _lib, _host

__all__ = ['uϕrot']


def uϕrot(U: Tensor, Φ: Tensor, Vi: Tensor) -> Tensor:
    r"""Rotate ``Vi`` about axis ``U`` by ``Φ`` (``utils.py:333-359``).

    ``Vo = cosΦ·Vi + (1-cosΦ)·(U·Vi)·U + sinΦ·U×Vi``

    Inputs:
        - ``U``:  `(N, *Nd, xyz)`, rotation axes, assumed unit;
        - ``Φ``:  `(N, *Nd,)`, rotation angles;
        - ``Vi``: `(N, *Nd, xyz, (nV))`, vectors to be rotated.
    Outputs:
        - ``Vo``: `(N, *Nd, xyz, (nV))`.
    """
    _host.require_device_tensor(U, 'U')
    _host.require_device_tensor(Vi, 'Vi')
    lib = _lib.require_library()
    dtype, device = Vi.dtype, Vi.device
    has_nv = Vi.ndim == U.ndim + 1
    assert has_nv or Vi.shape == U.shape
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
    return Vo
