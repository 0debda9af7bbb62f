r"""Host-side argument plumbing shared by the three entry points: dtype codes, flattening
``(N, *Nd)`` to the compact ``(N, nM)`` spin axis, and turning the reference's broadcastable
parameters (``()``, ``(N|1, *Nd|1)``, stride-0 expanded views) into (pointer, stride_n,
stride_m) descriptors without materialising them.
"""
from math import prod

import torch

from . import _lib

_SUPPORTED = (torch.float32, torch.float64)
_HALF = (torch.float16, torch.bfloat16)


def half_via_float(fn):
    r"""Decorator for the public entry points: fp16 / bf16 tensors -- which the reference accepts
    and integrates at that precision -- are computed in fp32 (every half tensor argument is
    upcast, autograd flows through the casts) and fp32 results are returned in the caller's half
    dtype.  Closer to exact arithmetic than the reference's half run, never bit-identical to it."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        is_half = lambda a: isinstance(a, torch.Tensor) and a.dtype in _HALF  # noqa: E731
        halfs = [a for a in list(args) + list(kw.values()) if is_half(a)]
        if not halfs:
            return fn(*args, **kw)
        out_dtype = halfs[0].dtype
        up = lambda a: a.float() if is_half(a) else a  # noqa: E731
        res = fn(*[up(a) for a in args], **{k: up(v) for k, v in kw.items()})
        down = lambda r: (r.to(out_dtype)  # noqa: E731
                          if isinstance(r, torch.Tensor) and r.dtype == torch.float32 else r)
        return tuple(down(r) for r in res) if isinstance(res, tuple) else down(res)
    return wrapper


def require_device_tensor(x: torch.Tensor, name: str):
    r"""The HIP path only: no CPU fallback exists (and none is wanted)."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"mrphy_amd: `{name}` must be a torch.Tensor")
    if x.device.type != 'cuda':
        raise RuntimeError(
            f"mrphy_amd: `{name}` is on {x.device}; this package only runs HIP kernels on a "
            "ROCm device tensor ('cuda:N'). There is no CPU fallback.")
    if x.dtype not in _SUPPORTED:
        raise NotImplementedError(
            f"mrphy_amd: `{name}` has dtype {x.dtype}; float32 and float64 are implemented")


def current_stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---------------------------------------------------------------------------------------------
# Precision of the fp32 time stepping (include/mrphy_hip.h: MRPHY_F32 vs MRPHY_F32P).
#   'precise' (default): S(phi^2), C(phi^2) evaluated in fp64 and rounded once, rounding errors of
#       the update carried.  4.8e-6 from exact arithmetic on the headline workload (128^3 x 4096
#       steps of up to 2.6 rad): inside the north star's 1e-5 at every pulse length tested.
#   'fast': the all-fp32 step (2.0e-5 there; the reference's own fp32 runs: 2.6-2.9e-5).  About
#       1.8x less arithmetic per step: matters for the VALU-bound fused kernels, not for the
#       HBM-bound blochsim over a materialised Beff.
# Both are deterministic and each is bit-identical between the fused and the two-kernel path.
# ---------------------------------------------------------------------------------------------
import contextvars as _cv
import os as _os

_PRECISION_DEFAULT = _os.environ.get('MRPHY_PRECISION', 'precise').lower()      # the process default
if _PRECISION_DEFAULT not in ('precise', 'fast'):
    raise ValueError(f"MRPHY_PRECISION must be 'precise' or 'fast', not {_PRECISION_DEFAULT!r}")
# ``with precision(...)`` is CONTEXT-LOCAL (contextvars: per thread, per asyncio task): two threads issuing forwards
# under different modes do not see each other's (round 4 kept a module global).  A new thread starts from the process
# default, as contextvars prescribe; a backward pass does not look here at all (its forward's dtype code is in ctx).
_PRECISION_CTX = _cv.ContextVar('mrphy_amd_precision', default=None)


class precision:
    r"""``with mrphy_amd.precision('fast'): ...`` selects the fp32 step arithmetic ('precise' |
    'fast', see above) for the calls of this thread / context inside the block; ``mrphy_amd.precision.get()`` is the
    mode in effect here, ``.set(mode)`` changes the PROCESS default (initially ``$MRPHY_PRECISION`` or 'precise').
    A backward pass uses the mode its forward ran in."""

    def __init__(self, mode: str):
        assert mode in ('precise', 'fast'), mode
        self.mode = mode
        self._tokens = []

    def __enter__(self):
        self._tokens.append(_PRECISION_CTX.set(self.mode))
        return self

    def __exit__(self, *exc):
        _PRECISION_CTX.reset(self._tokens.pop())
        return False

    @staticmethod
    def get() -> str:
        m = _PRECISION_CTX.get()
        return _PRECISION_DEFAULT if m is None else m

    @staticmethod
    def set(mode: str):
        global _PRECISION_DEFAULT
        assert mode in ('precise', 'fast'), mode
        _PRECISION_DEFAULT = mode


def dtype_code(data: torch.dtype, const: torch.dtype) -> int:
    if data == torch.float64:
        return _lib.F64
    if precision.get() == 'precise':
        return _lib.F32P_C64 if const == torch.float64 else _lib.F32P
    return _lib.F32_C64 if const == torch.float64 else _lib.F32


def pad_trailing(x: torch.Tensor, ndim: int) -> torch.Tensor:
    r"""Append singleton dims up to rank ``ndim`` (the reference right-pads, sims.py:309-313)."""
    return x.reshape(tuple(x.shape) + (ndim - x.ndim) * (1,))


class Bcast:
    r"""A per-spin constant broadcastable to ``(N, nM)``, kept alive with its strides."""
    __slots__ = ('t', 'sn', 'sm', '_nz')

    def __init__(self, x: torch.Tensor, N: int, Nd: tuple, dtype: torch.dtype,
                 device: torch.device):
        # x: shape (N|1, *Nd|1...) possibly with extra trailing singleton dims, or 0-dim
        x = x.to(device=device, dtype=dtype)
        lead = 1 + len(Nd)
        if x.ndim > lead:
            assert all(d == 1 for d in x.shape[lead:]), \
                f"cannot broadcast shape {tuple(x.shape)} over spins {(N,) + tuple(Nd)}"
            x = x.reshape(x.shape[:lead])
        x = pad_trailing(x, lead)
        assert x.shape[0] in (1, N) and all(a in (1, b) for a, b in zip(x.shape[1:], Nd)), \
            f"cannot broadcast shape {tuple(x.shape)} over spins {(N,) + tuple(Nd)}"
        sn = x.stride(0) if (x.shape[0] == N and N > 1) else 0
        spatial = tuple(x.shape[1:])
        if all(d == 1 for d in spatial):
            sm, t = 0, x                      # uniform over spins
        elif len(Nd) == 1:
            sm, t = x.stride(1), x            # compact layout: a plain stride (may be 0)
        else:
            # general *Nd: flatten; expanded views that cannot be viewed flat are copied (small)
            t = x.expand((x.shape[0],) + tuple(Nd)).reshape(x.shape[0], prod(Nd))
            sn = t.stride(0) if (t.shape[0] == N and N > 1) else 0
            sm = t.stride(1)
        self.t, self.sn, self.sm, self._nz = t, sn, sm, None

    @property
    def args(self):
        return (self.t.data_ptr(), self.sn, self.sm)

    def all_nonzero(self) -> bool:
        r"""No element is zero (one device-to-host read, cached on this object -- which is itself cached per
        constant tensor by ``sims._prep_constants``)."""
        if self._nz is None:
            self._nz = bool((self.t != 0).all())
        return self._nz


NULL_BC = (None, 0, 0)


def require_invertible_relaxation(code: int, e1, e2, who: str):
    r"""The precise fp32 adjoint (dtype codes F32P / F32P_C64) carries ``t = E h`` and divides by ``E`` once at the
    end (``csrc/bloch_math.hpp``: adj_end), as the reference's adjoint does at every step (``sims.py:174-177``): a
    relaxation factor that has underflowed to zero -- ``T2 < dt / 100`` in fp32 -- would turn the gradients into
    NaN.  Say so instead."""
    from . import _lib
    if code in (_lib.F32P, _lib.F32P_C64) and e1 is not None and e2 is not None \
            and not (e1.all_nonzero() and e2.all_nonzero()):
        raise RuntimeError(
            f"mrphy_amd.{who}: exp(-dt/T1) or exp(-dt/T2) is exactly 0 for some spin (T < dt/100 in fp32); the "
            "precise adjoint divides by it (as the reference's does, sims.py:174-177).  Use "
            "mrphy_amd.precision('fast') for such spins, fp64 data, or larger T1 / T2")


# ---------------------------------------------------------------------------------------------
# How the per-spin constants γ2πdt, E1 = exp(-dt/T1), E2, E1-1 are formed.
#
# The reference forms them with torch ops on the tensors' own device (sims.py:62,74-76).  They are
# per-spin CONSTANTS applied nT times, and exp() is not bit-reproducible: ROCm's fp32 exp differs
# from torch's vectorised CPU one by 1 ulp on 12.7 % of the spins of the synthetic cube (and two CPUs
# differ too).  One ulp of E times nT steps is nT * 6e-8 on the affected spins: 2.1e-5 relative L2 of
# Mo at nT = 1024 -- more than the arithmetic error of the integration.  Three modes:
#
#   default ('rounded')   γ2πdt and the argument q = -dt/T by the reference's own expressions
#                         (plain IEEE arithmetic: the same bits on any device); E = exp(q) evaluated
#                         in fp64 and rounded ONCE to the constants' dtype -- the correctly rounded
#                         exp of the reference's own argument.  Device-independent and reproducible
#                         across boxes (CPU and GPU fp64 exp agree to an fp64 ulp; a difference
#                         survives the rounding for ~1 spin in 10^8); at most 1 ulp from ANY platform's
#                         exp, and from torch's CPU exp on 6 % of the spins instead of 12.7 %.
#   constants_on('cpu')   the reference's expressions, torch.exp in the data dtype, on the CPU: the
#   (or any device)       very constants of a reference run on that device.  The parity tests use this
#                         against the CPU oracle / the golden vectors: what is compared is then the
#                         kernels' arithmetic, not two exp() implementations.
#   constants_on('native') as above on the inputs' own device (the reference's literal behaviour;
#                         this package's default until round 3).
# The kernels are identical in all three.
# ---------------------------------------------------------------------------------------------
_CONST_UNSET = object()
_CONST_CTX = _cv.ContextVar('mrphy_amd_constants_on', default=_CONST_UNSET)     # None: 'rounded'; 'native'; or a torch.device


def _const_mode():
    m = _CONST_CTX.get()
    return None if m is _CONST_UNSET else m


class constants_on:
    r"""``with constants_on('cpu'): ...`` -- form γ2πdt, E1, E2, E1-1 with torch's own ops on that
    device (``'native'``: on the inputs' device); ``None`` restores the default (``exp`` evaluated in
    fp64, rounded once: see above).  Context-local like :class:`precision`."""

    def __init__(self, device):
        self.mode = device if device in (None, 'native') else torch.device(device)
        self._tokens = []

    def __enter__(self):
        self._tokens.append(_CONST_CTX.set(self.mode))
        return self

    def __exit__(self, *exc):
        _CONST_CTX.reset(self._tokens.pop())
        return False


def const_device(default: torch.device) -> torch.device:
    r"""Device the constants are formed on."""
    m = _const_mode()
    return m if isinstance(m, torch.device) else default


def const_exp_rounded_once() -> bool:
    r"""True in the default mode: ``exp`` in fp64, one rounding."""
    return _const_mode() is None


def const_mode_key() -> str:
    m = _const_mode()
    return 'rounded' if m is None else str(m)
