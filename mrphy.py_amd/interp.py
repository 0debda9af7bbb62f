r"""On-device multi-scale time resampling of a pulse: ``mrphy.mobjs.Pulse.interpT`` (reference
``mrphy/mobjs.py:177-220``) for every ``kind`` of ``scipy.interpolate.interp1d`` that the
reference's ``kind`` argument reaches (``mobjs.py:201,214-215``): ``'linear'`` and the one-tap kinds
(``'nearest'``, ``'nearest-up'``, ``'previous'``, ``'next'``, ``'zero'``) through the HIP kernels, the
spline kinds (``'slinear'``, ``'quadratic'``, ``'cubic'``, integer orders) through their interpolation
operator, which scipy itself supplies once per grid (:func:`interp_matrix`).

The reference detaches the waveforms, copies them to the host, interpolates with
``scipy.interpolate.interp1d`` and builds a new ``Pulse`` -- a device->host->device round trip per
resampling that also cuts the autograd graph.  Here only the resampling GRID is formed on the host
(it depends on ``nT``, ``dt`` and the new ``dt`` alone and reproduces the reference's float64
arithmetic, including its ``//`` floor of the sample count, ``mobjs.py:211-212``); the waveforms
stay on the device and the map is differentiable (the adjoint is the transposed scatter).
"""
from typing import Tuple

import numpy as np
import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host

__all__ = ['interpT', 'interp_grid', 'interp_select', 'interp_matrix', 'SELECT_KINDS', 'SPLINE_KINDS']

SELECT_KINDS = ('nearest', 'nearest-up', 'previous', 'next', 'zero')
SPLINE_KINDS = ('slinear', 'quadratic', 'cubic')       # and integer spline orders 1..5
_MATRIX_MAX = 1 << 26      # elements of the dense spline operator we are willing to hold (512 MB fp64)

_grid_cache = {}


def interp_grid(nT: int, dt_old: float, dt_new: float):
    r"""``lo, w, dx`` (and the new sample count) for resampling ``nT`` samples of dwell ``dt_old``
    to dwell ``dt_new``, as ``Pulse.interpT`` + ``interp1d(kind='linear', assume_sorted=True)``
    compute them: source times ``t_o = arange(nT+1)*dt_old`` (a zero sample is prepended,
    ``mobjs.py:204-207``), new times ``t_n = arange(1, t_o[-1]//dt_new + 1)*dt_new``."""
    t_o = np.arange(0, nT + 1) * dt_old
    t_n = np.arange(1, t_o[-1] // dt_new + 1) * dt_new
    hi = np.searchsorted(t_o, t_n).clip(1, len(t_o) - 1).astype(np.int64)
    lo = hi - 1
    return lo.astype(np.int32), t_n - t_o[lo], t_o[hi] - t_o[lo], len(t_n)


def interp_select(nT: int, dt_old: float, dt_new: float, kind: str):
    r"""``sel`` (int32, one entry per new sample) and the new sample count for the one-tap kinds:
    ``sel[j]`` is the index into the ZERO-PREPENDED source (0 = the prepended sample, ``k >= 1`` =
    sample ``k - 1`` of the pulse) that ``interp1d(t_o, ., kind=kind, assume_sorted=True)(t_n)``
    returns for new sample ``j``, on the grid ``Pulse.interpT`` builds (``mobjs.py:209-212``).

    The index is taken from scipy itself -- the routine the reference calls -- by resampling the
    index ramp ``0, 1, 2, ...``: a one-tap kind returns exactly the index it selected, so ties,
    knots and the ends follow scipy's conventions for every kind (``'zero'`` goes through its
    order-0 B-spline, the others through ``searchsorted``) without restating them.  The grid
    depends on ``(nT, dt_old, dt_new, kind)`` only; the waveform never leaves the device.
    """
    if kind not in SELECT_KINDS:
        raise ValueError(f"interp_select: kind must be one of {SELECT_KINDS}, not {kind!r}")
    try:
        from scipy import interpolate
    except ImportError as e:                       # the reference itself requires scipy (setup.py:11)
        raise ImportError(f"mrphy_amd.interp: kind={kind!r} takes its resampling index from "
                          "scipy.interpolate.interp1d (as the reference does); scipy is missing") from e
    t_o = np.arange(0, nT + 1) * dt_old
    t_n = np.arange(1, t_o[-1] // dt_new + 1) * dt_new
    ramp = np.arange(nT + 1, dtype=np.float64)
    got = interpolate.interp1d(t_o, ramp, kind=kind, copy=False, assume_sorted=True)(t_n) \
        if len(t_n) else np.zeros(0)
    sel = np.rint(got).astype(np.int64)
    # These guard a kernel launch (the kernel trusts the range of sel): real errors, not asserts.
    if not np.array_equal(sel.astype(np.float64), got):
        raise RuntimeError(f"interp_select: kind={kind!r} did not return sample indices")
    if sel.size and (sel.min() < 0 or sel.max() > nT or np.any(np.diff(sel) < 0)):
        raise RuntimeError(f"interp_select: kind={kind!r} gave indices outside [0, {nT}] or not "
                           "non-decreasing")
    return sel.astype(np.int32), len(t_n)


def _is_spline(kind) -> bool:
    return kind in SPLINE_KINDS or (isinstance(kind, int) and not isinstance(kind, bool) and 1 <= kind <= 5)


def interp_matrix(nT: int, dt_old: float, dt_new: float, kind):
    r"""The resampling operator ``W`` `(nTn, nT)` (float64) of ``Pulse.interpT(..., kind=kind)`` for
    scipy's spline kinds: new waveform = ``W @ old waveform`` along time.

    ``interp1d`` with a spline kind is a LINEAR map of the samples (a global B-spline collocation
    solve -- not-a-knot ends for ``'quadratic'``/``'cubic'`` -- followed by evaluation), so scipy,
    applied once to the unit vectors on the grid ``Pulse.interpT`` builds (``mobjs.py:204-212``: a zero
    sample prepended, ``t_n = arange(1, t_o[-1]//dt_new + 1)*dt_new``), returns the operator itself;
    nothing of scipy's knot/boundary conventions is restated.  The column of the prepended sample is
    dropped (it multiplies the zero).  Entries decay like 0.27^|i-j| (cubic) away from the diagonal, so
    the dense form is mostly zeros; it is O(nT^2) memory, once per grid (refused beyond 2^26 entries)."""
    if not _is_spline(kind):
        raise ValueError(f"interp_matrix: kind must be one of {SPLINE_KINDS} or an integer spline "
                         f"order 1..5, not {kind!r}")
    try:
        from scipy import interpolate
    except ImportError as e:                       # the reference itself requires scipy (setup.py:11)
        raise ImportError(f"mrphy_amd.interp: kind={kind!r} takes its operator from "
                          "scipy.interpolate.interp1d (as the reference does); scipy is missing") from e
    t_o = np.arange(0, nT + 1) * dt_old
    t_n = np.arange(1, t_o[-1] // dt_new + 1) * dt_new
    if (nT + 1) * max(len(t_n), nT + 1) > _MATRIX_MAX:
        raise NotImplementedError(
            f"mrphy_amd.interp: kind={kind!r} for nT = {nT} -> {len(t_n)} samples would need a dense "
            f"{len(t_n)} x {nT + 1} operator; use 'linear' or a one-tap kind at this length")
    if len(t_n) == 0:
        return np.zeros((0, nT)), 0
    W = interpolate.interp1d(t_o, np.eye(nT + 1), axis=0, kind=kind, copy=False, assume_sorted=True)(t_n)
    return np.ascontiguousarray(W[:, 1:]), len(t_n)


class _InterpSelectHIP(Function):
    r"""``out = _InterpSelectHIP.apply(y, sel, nTn)``: ``mrphy_pulse_interp_select`` forward and
    adjoint (the transposed selection, a gather-sum)."""

    @staticmethod
    def forward(ctx, y, sel, nTn):
        lib = _lib.require_library()
        yc = y.detach().contiguous()
        nch, nTo = int(np.prod(yc.shape[:-1])), yc.shape[-1]
        out = torch.empty(yc.shape[:-1] + (nTn,), dtype=yc.dtype, device=yc.device)
        code = _lib.F64 if yc.dtype == torch.float64 else _lib.F32
        with torch.cuda.device(yc.device):
            rc = lib.mrphy_pulse_interp_select(code, 1, yc.data_ptr(), out.data_ptr(), sel.data_ptr(),
                                               nch, nTo, nTn, _host.current_stream(yc.device))
        _lib.check(rc, 'mrphy_pulse_interp_select')
        ctx.save_for_backward(sel)
        ctx.dims = (nch, nTo, nTn, code, y.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_library()
        (sel,) = ctx.saved_tensors
        nch, nTo, nTn, code, shape = ctx.dims
        gc = g.contiguous()
        gy = torch.empty(shape, dtype=gc.dtype, device=gc.device)
        with torch.cuda.device(gc.device):
            rc = lib.mrphy_pulse_interp_select(code, -1, gc.data_ptr(), gy.data_ptr(), sel.data_ptr(),
                                               nch, nTo, nTn, _host.current_stream(gc.device))
        _lib.check(rc, 'mrphy_pulse_interp_select (adjoint)')
        return gy, None, None


class _InterpLinearHIP(Function):
    @staticmethod
    def forward(ctx, y, lo, w, dx, nTn):
        lib = _lib.require_library()
        yc = y.detach().contiguous()
        nch, nTo = int(np.prod(yc.shape[:-1])), yc.shape[-1]
        out = torch.empty(yc.shape[:-1] + (nTn,), dtype=yc.dtype, device=yc.device)
        code = _lib.F64 if yc.dtype == torch.float64 else _lib.F32
        with torch.cuda.device(yc.device):
            rc = lib.mrphy_pulse_interp_linear(code, 1, yc.data_ptr(), out.data_ptr(), lo.data_ptr(),
                                               w.data_ptr(), dx.data_ptr(), nch, nTo, nTn,
                                               _host.current_stream(yc.device))
        _lib.check(rc, 'mrphy_pulse_interp_linear')
        ctx.save_for_backward(lo, w, dx)
        ctx.dims = (nch, nTo, nTn, code, y.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_library()
        lo, w, dx = ctx.saved_tensors
        nch, nTo, nTn, code, shape = ctx.dims
        gc = g.contiguous()
        gy = torch.empty(shape, dtype=gc.dtype, device=gc.device)
        with torch.cuda.device(gc.device):
            rc = lib.mrphy_pulse_interp_linear(code, -1, gc.data_ptr(), gy.data_ptr(), lo.data_ptr(),
                                               w.data_ptr(), dx.data_ptr(), nch, nTo, nTn,
                                               _host.current_stream(gc.device))
        _lib.check(rc, 'mrphy_pulse_interp_linear (adjoint)')
        return gy, None, None, None, None


def interpT(rf: Tensor, gr: Tensor, dt: Tensor, dt_new: Tensor, *, kind: str = 'linear'
            ) -> Tuple[Tensor, Tensor, Tensor]:
    r"""Resample a pulse ``rf (N,xy,nT,(nCoils))``, ``gr (N,xyz,nT)`` of dwell ``dt`` to dwell
    ``dt_new`` -- what ``Pulse.interpT(dt_new, kind=kind)`` returns as ``(rf, gr, dt)`` of the new
    pulse (``mobjs.py:177-220``), computed on the device and differentiable w.r.t. ``rf``/``gr``.
    ``kind``: ``'linear'`` (default), one of :data:`SELECT_KINDS`, or a spline kind
    (:data:`SPLINE_KINDS`, integer orders 1..5): those apply scipy's own interpolation operator for the
    grid (:func:`interp_matrix`, cached on the device) as one fp64 matrix product, rounded once to the
    waveform dtype -- what scipy computes (in fp64) and ``Pulse.interpT`` rounds (``mobjs.py:217``).

    As in the reference both ``dt`` and ``dt_new`` must hold a single value (``mobjs.py:193``);
    equal dwell times return the inputs unchanged.
    """
    spline = _is_spline(kind)
    if kind != 'linear' and kind not in SELECT_KINDS and not spline:
        raise NotImplementedError(
            f"mrphy_amd.interp.interpT: kind={kind!r} is not a kind of scipy.interpolate.interp1d that "
            "this package knows; available: 'linear', "
            + ', '.join(repr(k) for k in SELECT_KINDS + SPLINE_KINDS) + ', integer spline orders 1..5')
    assert dt.numel() == dt_new.numel() == 1
    _host.require_device_tensor(rf, 'rf')
    _host.require_device_tensor(gr, 'gr')
    nT, dev = rf.shape[2], rf.device
    # The grid depends on (nT, dt, dt_new) only.  Reading the two dwell times is a device->host
    # sync; in a multi-scale design loop they are the same tensors every iteration, so the grid
    # (and its device copies) is cached per (identity, version) of dt and dt_new.
    from . import sims
    key = ((sims._tkey(dt), sims._tkey(dt_new)), nT, str(dev), kind)
    hit = sims._cache_get(_grid_cache, key, (dt, dt_new))
    if hit is None:
        dt_o, dt_n = dt.item(), dt_new.item()
        if dt_o == dt_n:
            hit = ()                               # equal dwell times: nothing to resample
        else:
            # the new pulse's dt: ONE tensor per cache entry, so that what is keyed on it
            # downstream (the relaxation constants) stays cached as well
            dt_out = dt_new.detach().to(device=dev, dtype=rf.dtype).reshape(dt_new.shape)
            if kind == 'linear':
                lo, w, dx, nTn = interp_grid(nT, dt_o, dt_n)
                grid = (torch.from_numpy(lo).to(dev), torch.from_numpy(w).to(dev),
                        torch.from_numpy(dx).to(dev))
            elif spline:
                W, nTn = interp_matrix(nT, dt_o, dt_n, kind)
                grid = (torch.from_numpy(W).to(dev),)
            else:
                sel, nTn = interp_select(nT, dt_o, dt_n, kind)
                grid = (torch.from_numpy(sel).to(dev),)
            hit = (grid, nTn, dt_out)
        sims._cache_put(_grid_cache, key, (dt, dt_new), hit)
    if not hit:
        return rf, gr, dt
    grid, nTn, dt_out = hit
    if spline:
        # one fp64 GEMM with the cached operator (plain torch ops: autograd supplies W^T g); rounded
        # once to the waveform dtype, as Pulse.interpT rounds scipy's fp64 result
        resample = lambda y: torch.matmul(y.to(torch.float64), grid[0].T).to(y.dtype)  # noqa: E731
    else:
        op = _InterpLinearHIP if kind == 'linear' else _InterpSelectHIP
        resample = lambda y: op.apply(y, *grid, nTn)  # noqa: E731
    if rf.ndim == 4:                               # (N, xy, nT, nC): time is not the last axis
        rf_n = resample(rf.movedim(2, -1)).movedim(-1, 2)
    else:
        rf_n = resample(rf)
    return rf_n, resample(gr), dt_out
