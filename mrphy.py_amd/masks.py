r"""Mask gather/scatter and cube locations: the steps either side of the hot path in
``mobjs.SpinArray.applypulse`` (reference ``mobjs.py:427-433,449``).

The reference moves between the spatial layout `(N, *Nd, ...)` and the compact one `(N, nM, ...)`
with boolean-mask indexing (``v[mask]``, ``out[mask] = v_``; ``mobjs.py:512-553``), which counts
the mask on the host at every call.  Here a mask is turned once into a :class:`MaskIndex` (two
int32 lists on the device) and the gather/scatter run as HIP kernels with no host round trip:

    ``index = MaskIndex(mask)``                   # once per mask (one count on the host)
    ``v_ = extract(v, index)``                    # SpinArray.extract
    ``v = embed(v_, index)``                      # SpinArray.embed  (NaN outside the mask)
    ``loc_ = cube_loc(index, fov, ofst)``         # SpinCube._update_loc_ (mobjs.py:815-839)

``extract`` and ``embed`` are each other's adjoints and are differentiable (the reference's
indexing is, too).
"""
import struct
from math import prod
from typing import Optional, Union

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host

__all__ = ['MaskIndex', 'extract', 'embed', 'cube_loc']

_NAN_BITS = {torch.float32: struct.unpack('<I', struct.pack('<f', float('nan')))[0],
             torch.float64: struct.unpack('<Q', struct.pack('<d', float('nan')))[0]}


class MaskIndex:
    r"""Index lists of a ``SpinArray`` mask (``mobjs.py:253,289``: `(1, *Nd)` bool, one mask for
    the whole batch).

    Attributes:
        - ``Nd``:  spatial shape; ``nV = prod(Nd)``; ``nM``: number of spins in the mask;
        - ``idx``: `(nM,)` int32, voxel number (row-major over ``Nd``) of compact spin ``j``;
        - ``inv``: `(nV,)` int32, compact spin number of voxel ``p``, −1 outside the mask
          (the map ``mobjs.py:493-497`` builds for cropping).
    """
    __slots__ = ('Nd', 'nV', 'nM', 'idx', 'inv', 'device')

    def __init__(self, mask: Tensor):
        assert mask.dtype == torch.bool and mask.ndim >= 2 and mask.shape[0] == 1, \
            "mask must be a (1, *Nd) bool tensor"
        if mask.device.type != 'cuda':
            raise RuntimeError("mrphy_amd: `mask` must live on the ROCm device ('cuda:N'); "
                               "there is no CPU fallback")
        self.device = mask.device
        self.Nd = tuple(mask.shape[1:])
        self.nV = prod(self.Nd)
        assert self.nV < 2 ** 31, "spatial grids of 2^31 voxels or more are not supported"
        flat = mask.reshape(-1)
        self.idx = torch.nonzero(flat).reshape(-1).to(torch.int32)    # the one host sync
        self.nM = int(self.idx.numel())
        inv = torch.full((self.nV,), -1, dtype=torch.int32, device=mask.device)
        inv[self.idx.long()] = torch.arange(self.nM, dtype=torch.int32, device=mask.device)
        self.inv = inv


def _as_index(m: Union[Tensor, MaskIndex]) -> MaskIndex:
    return m if isinstance(m, MaskIndex) else MaskIndex(m)


def _launch_extract(v: Tensor, ix: MaskIndex, out_: Tensor, N: int, K: int):
    lib = _lib.require_library()
    with torch.cuda.device(v.device):
        rc = lib.mrphy_mask_extract(v.element_size(), v.data_ptr(), ix.idx.data_ptr(),
                                    out_.data_ptr(), N, ix.nV, ix.nM, K,
                                    _host.current_stream(v.device))
    _lib.check(rc, 'mrphy_mask_extract')


def _launch_embed(v_: Tensor, ix: MaskIndex, out: Tensor, N: int, K: int, fill: Optional[int]):
    lib = _lib.require_library()
    with torch.cuda.device(out.device):
        rc = lib.mrphy_mask_embed(out.element_size(), v_.data_ptr(), ix.inv.data_ptr(),
                                  out.data_ptr(), N, ix.nV, ix.nM, K,
                                  0 if fill is None else 1, 0 if fill is None else fill,
                                  _host.current_stream(out.device))
    _lib.check(rc, 'mrphy_mask_embed')


class _Extract(Function):
    @staticmethod
    def forward(ctx, v, ix):
        N, tail = v.shape[0], tuple(v.shape[1 + len(ix.Nd):])
        vc = v.detach().contiguous()
        out_ = vc.new_empty((N, ix.nM) + tail)
        _launch_extract(vc, ix, out_, N, prod(tail))
        ctx.ix = ix
        return out_

    @staticmethod
    def backward(ctx, g_):
        ix = ctx.ix
        N, tail = g_.shape[0], tuple(g_.shape[2:])
        g = g_.new_empty((N,) + ix.Nd + tail)
        _launch_embed(g_.contiguous(), ix, g, N, prod(tail), 0)       # zeros outside the mask
        return g, None


class _Embed(Function):
    @staticmethod
    def forward(ctx, v_, ix):
        N, tail = v_.shape[0], tuple(v_.shape[2:])
        vc = v_.detach().contiguous()
        out = vc.new_empty((N,) + ix.Nd + tail)
        _launch_embed(vc, ix, out, N, prod(tail), _NAN_BITS[vc.dtype])
        ctx.ix = ix
        return out

    @staticmethod
    def backward(ctx, g):
        ix = ctx.ix
        N, tail = g.shape[0], tuple(g.shape[1 + len(ix.Nd):])
        gc = g.contiguous()
        g_ = gc.new_empty((N, ix.nM) + tail)
        _launch_extract(gc, ix, g_, N, prod(tail))
        return g_, None


def extract(v: Tensor, mask: Union[Tensor, MaskIndex], *, out_: Optional[Tensor] = None
            ) -> Tensor:
    r"""``SpinArray.extract`` (``mobjs.py:532-553``): keep the voxels of the mask, compactly.

    Inputs:
        - ``v``: `(N, *Nd, ...)`;
        - ``mask``: `(1, *Nd)` bool, or the :class:`MaskIndex` made from it (reuse it).
    Optionals:
        - ``out_``: `(N, nM, ...)`, in-place holder, must be contiguous.
    Outputs:
        - ``out_``: `(N, nM, ...)`.
    """
    _host.require_device_tensor(v, 'v')
    ix = _as_index(mask)
    nd = len(ix.Nd)
    assert tuple(v.shape[1:1 + nd]) == ix.Nd, \
        f"`v` {tuple(v.shape)} does not carry the mask's spatial shape {ix.Nd}"
    if out_ is None:
        return _Extract.apply(v, ix)
    N, tail = v.shape[0], tuple(v.shape[1 + nd:])
    assert out_.is_contiguous() and tuple(out_.shape) == (N, ix.nM) + tail \
        and out_.dtype == v.dtype and out_.device == v.device
    if torch.is_grad_enabled() and v.requires_grad:
        return out_.copy_(_Extract.apply(v, ix))
    _launch_extract(v.detach().contiguous(), ix, out_, N, prod(tail))
    return out_


def embed(v_: Tensor, mask: Union[Tensor, MaskIndex], *, out: Optional[Tensor] = None) -> Tensor:
    r"""``SpinArray.embed`` (``mobjs.py:512-530``): put compact data back on the grid.

    Inputs:
        - ``v_``: `(N, nM, ...)`;
        - ``mask``: `(1, *Nd)` bool, or the :class:`MaskIndex` made from it.
    Optionals:
        - ``out``: `(N, *Nd, ...)`, in-place holder (contiguous); voxels outside the mask keep
          their values.  Without it a new tensor is returned, NaN outside the mask.
    Outputs:
        - ``out``: `(N, *Nd, ...)`.
    """
    _host.require_device_tensor(v_, 'v_')
    ix = _as_index(mask)
    assert v_.shape[1] == ix.nM, f"`v_` {tuple(v_.shape)} does not have nM = {ix.nM} spins"
    if out is None:
        return _Embed.apply(v_, ix)
    N, tail = v_.shape[0], tuple(v_.shape[2:])
    assert out.is_contiguous() and tuple(out.shape) == (N,) + ix.Nd + tail \
        and out.dtype == v_.dtype and out.device == v_.device
    if torch.is_grad_enabled() and v_.requires_grad:
        # differentiable in-place form, as the reference's ``out[mask] = v_``
        fresh = _Embed.apply(v_, ix)
        inside = (ix.inv >= 0).reshape((1,) + ix.Nd + (1,) * len(tail))
        return out.copy_(torch.where(inside, fresh, out))
    _launch_embed(v_.detach().contiguous(), ix, out, N, prod(tail), None)
    return out


def cube_loc(mask: Union[Tensor, MaskIndex], fov: Tensor, ofst: Tensor, *,
             out_: Optional[Tensor] = None) -> Tensor:
    r"""``SpinCube._update_loc_`` (``mobjs.py:815-839``): compact spin locations of a 3-D grid.

    ``loc_[n, j, i] = fov[n, i]·(c_i − Nd_i//2)/Nd_i + ofst[n, i]`` with ``c`` the grid
    subscripts of compact spin ``j``.

    Inputs:
        - ``mask``: `(1, nx, ny, nz)` bool or its :class:`MaskIndex`;
        - ``fov``, ``ofst``: `(N, xyz)`, "cm".
    Optionals:
        - ``out_``: `(N, nM, xyz)`, in-place holder, contiguous.
    Outputs:
        - ``loc_``: `(N, nM, xyz)`, "cm".
    """
    _host.require_device_tensor(fov, 'fov')
    ix = _as_index(mask)
    assert len(ix.Nd) == 3, "cube_loc is for 3-D grids (SpinCube)"
    assert fov.ndim == 2 and fov.shape[1] == 3 and ofst.shape == fov.shape
    N, dtype, device = fov.shape[0], fov.dtype, fov.device
    if out_ is None:
        out_ = torch.empty((N, ix.nM, 3), dtype=dtype, device=device)
    assert out_.is_contiguous() and tuple(out_.shape) == (N, ix.nM, 3) and out_.dtype == dtype
    lib = _lib.require_library()
    f, o = fov.detach().contiguous(), ofst.detach().to(device=device, dtype=dtype).contiguous()
    with torch.cuda.device(device):
        rc = lib.mrphy_cube_loc(_lib.F64 if dtype == torch.float64 else _lib.F32,
                                ix.idx.data_ptr(), f.data_ptr(), o.data_ptr(), out_.data_ptr(),
                                N, ix.nM, *ix.Nd, _host.current_stream(device))
    _lib.check(rc, 'mrphy_cube_loc')
    return out_
