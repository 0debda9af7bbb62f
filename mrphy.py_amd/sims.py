r"""Bloch simulation with explicit adjoint, HIP-backed.

Drop-in for ``mrphy.sims.blochsim`` (reference ``mrphy/sims.py:272-315``) and the autograd
pair ``BlochSim.forward/backward`` (``sims.py:32-132``, ``135-269``).  Same signature, same
layouts, same asserts; differentiable w.r.t. ``Mi`` and ``Beff`` only (``sims.py:27``).

What differs from the reference, on purpose:

* the time loop runs inside one kernel launch (``mrphy_blochsim_fwd``) instead of 14 ATen
  launches per step; history is kept only when a gradient is needed, and is 12 B per
  spin-step (the magnetisation before each step) instead of 40 B (``sims.py:84-88``);
* ``grad_Beff`` is a fresh tensor: the reference overwrites its saved ``γBeff``
  (``sims.py:239-264``), which breaks a second ``backward``;
* ``grad_Mi`` is right for per-spin ``γ`` and per-batch ``dt`` (the reference divides by
  ``γ2πdt[0, ...]``, ``sims.py:267``).
"""
import weakref
from math import pi as π, prod
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host, _hist
from ._consts import γH, dt0

__all__ = ['blochsim', 'blochsim_consts', 'freeprec']


def _gamma_dt_constants(T1, T2, γ, dt):
    r"""γ2πdt, E1, E2, E1-1 exactly as the reference forms them (``sims.py:62,74-76``):
    same expressions, hence the same dtype promotion and the same roundings -- except that, in
    the default constants mode, ``exp`` is evaluated in fp64 and rounded once (device-independent
    bits; ``_host.constants_on``).  Inputs are already padded to the rank of Beff; outputs keep it.
    """
    γ2πdt = 2*π*γ*dt
    if T1 is None:
        return γ2πdt, None, None, None
    q1, q2 = -dt/T1, -dt/T2
    if _host.const_exp_rounded_once() and q1.dtype != torch.float64:
        # default mode (see _host.constants_on): the correctly rounded exp of the reference's own
        # argument -- the same bits on every device; E1 - 1 stays the reference's fp32 subtraction
        E1, E2 = torch.exp(q1.double()).to(q1.dtype), torch.exp(q2.double()).to(q2.dtype)
    else:
        E1, E2 = torch.exp(q1), torch.exp(q2)
    return γ2πdt, E1, E2, E1 - 1


# ---------------------------------------------------------------------------------------------
# Small caches for optimisation loops, where T1, T2, γ, dt are the same tensors call after call:
# the constants (6-9 small torch launches) and their broadcast descriptors are about two thirds of
# the ~120 us a call costs on the host, and at 32^3-class problems the host is the bottleneck.
# A key holds (id, version) of each input tensor; a weak reference per input guards against a new
# tensor reusing a dead one's id, the version counter against in-place updates.
# ---------------------------------------------------------------------------------------------
_CACHE_MAX = 16
_const_cache = {}
_prep_cache = {}


_NOCACHE = 'nocache'


def _tkey(x):
    r"""(id, version) of a tensor; ``_NOCACHE`` for tensors without a version counter (created
    under ``torch.inference_mode()``): calls with such inputs are simply not cached."""
    if x is None:
        return None
    try:
        if x.is_inference():
            return _NOCACHE
        return (id(x), x._version)
    except RuntimeError:
        return _NOCACHE


def _cache_get(cache, key, tensors):
    if _NOCACHE in key[0]:
        return None
    hit = cache.get(key)
    if hit is None:
        return None
    refs, value = hit
    for r, x in zip(refs, tensors):
        if (r is None) != (x is None) or (r is not None and r() is not x):
            del cache[key]
            return None
    return value


def _cache_put(cache, key, tensors, value):
    if _NOCACHE in key[0]:
        return value
    if len(cache) >= _CACHE_MAX:
        cache.pop(next(iter(cache)))
    cache[key] = (tuple(None if x is None else weakref.ref(x) for x in tensors), value)
    return value


def relax_constants(T1, T2, γ, dt, ndim: int, device):
    r"""``γ2πdt, E1, E2, E1-1`` for ``Beff`` of rank ``ndim`` on ``device``: {γ, dt, T1, T2} padded to
    that rank by trailing singleton dims (``sims.py:309-313``), then the reference's own
    expressions (:func:`_gamma_dt_constants`) on the constants' device; cached per input tensors."""
    cdev = _host.const_device(device)
    ins = (T1, T2, γ, dt)
    key = (tuple(_tkey(x) for x in ins), ndim, str(cdev), _host.const_mode_key())
    hit = _cache_get(_const_cache, key, ins)
    if hit is not None:
        return hit
    pad = lambda x: None if x is None else _host.pad_trailing(x.detach().to(cdev), ndim)  # noqa: E731
    with torch.no_grad():
        out = _gamma_dt_constants(pad(T1), pad(T2), pad(γ), pad(dt))
    return _cache_put(_const_cache, key, ins, out)


def _prep_constants(γ2πdt, E1, E2, E1_1, N, Nd, data_dtype, device):
    r"""Common constant dtype + broadcast descriptors for the C ABI (cached per constant tensors)."""
    ins = (γ2πdt, E1, E2, E1_1)
    key = (tuple(_tkey(x) for x in ins), N, tuple(Nd), data_dtype, str(device), _host.precision.get())
    hit = _cache_get(_prep_cache, key, ins)
    if hit is not None:
        return hit
    return _cache_put(_prep_cache, key, ins, _prep_constants_uncached(*ins, N, Nd, data_dtype, device))


def _prep_constants_uncached(γ2πdt, E1, E2, E1_1, N, Nd, data_dtype, device):
    cs = [c for c in (γ2πdt, E1, E2, E1_1) if c is not None]
    wide = any(c.dtype == torch.float64 for c in cs)
    cdt = torch.float64 if (wide or data_dtype == torch.float64) else torch.float32
    code = _host.dtype_code(data_dtype, cdt)
    mk = lambda c: None if c is None else _host.Bcast(c, N, Nd, cdt, device)  # noqa: E731
    g, e1, e2 = mk(γ2πdt), mk(E1), mk(E2)
    e1m1 = mk(E1_1)
    if e1m1 is not None and (e1m1.sn, e1m1.sm) != (e1.sn, e1.sm):
        # E1-1 travels with E1's strides (mrphy_hip.h).  Constants formed together share them;
        # user-supplied ones (blochsim_consts) may not -- e.g. an expanded E1 beside a contiguous
        # E1_1: materialise both over (N, *Nd), as slowsims.blochsim_1step does.
        def dense(c):
            lead = 1 + len(Nd)
            x = c.to(device=device, dtype=cdt)
            if x.ndim > lead:
                x = x.reshape(x.shape[:lead])
            x = _host.pad_trailing(x, lead).expand((N,) + tuple(Nd)).contiguous()
            return _host.Bcast(x, N, Nd, cdt, device)
        e1, e1m1 = dense(E1), dense(E1_1)
    return code, g, e1, e2, e1m1


def _reduce_to_const(g_full: Tensor, c: Tensor, N: int, Nd: tuple) -> Tensor:
    r"""Gradient of a per-spin quantity `(N, *Nd)` w.r.t. a broadcast constant ``c`` (shape
    `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1, 1...)`, right-padded as at ``sims.py:309-313``): summed over the
    broadcast axes, in ``c``'s shape and dtype."""
    lead = 1 + len(Nd)
    shp = tuple(c.shape[:lead]) + (1,) * (lead - min(c.ndim, lead))
    return g_full.sum_to_size(shp).reshape(c.shape).to(c.dtype)


class BlochSimHIP(Function):
    r"""``Mo = BlochSimHIP.apply(Mi, Beff, γ2πdt, E1, E2, E1_1, need_hist)`` -- the kernels take
    the per-spin constants, however they were formed (see :func:`blochsim`,
    :func:`blochsim_consts`).  ``need_hist`` (keep the 12 B/spin-step history for the adjoint) is
    decided by the caller from grad mode and ``requires_grad``: ``ctx.needs_input_grad`` stays
    ``True`` under ``torch.no_grad()``.

    Constants that require grad (round 3; reached through ``slowsims`` and :func:`blochsim_consts`,
    whose reference counterparts are plain differentiable torch ops) get their gradients from the same
    backward sweep (``mrphy_blochsim_bwd_consts``): per spin, then summed over the axes the constant
    broadcasts along."""

    @staticmethod
    def forward(ctx, Mi: Tensor, Beff: Tensor, γ2πdt: Tensor, E1: Optional[Tensor],
                E2: Optional[Tensor], E1_1: Optional[Tensor], need_hist: bool = False, ws=None) -> Tensor:
        lib = _lib.require_library()
        device, dtype = Mi.device, Mi.dtype
        NNd, nT = tuple(Beff.shape[:-2]), Beff.shape[-2]
        N, Nd = NNd[0], NNd[1:]
        nM = prod(Nd)
        code, g, e1, e2, e1m1 = _prep_constants(γ2πdt, E1, E2, E1_1, N, Nd, dtype, device)

        Mi_c = Mi.detach().contiguous()
        Beff_c = Beff.detach().to(dtype).contiguous()
        Mo = torch.empty(NNd + (3,), dtype=dtype, device=device)
        need_hist = bool(need_hist)
        # history for the adjoint: opaque buffer(s) in the library's own (tile-SoA) layout
        hist = None
        if need_hist:
            if ws is not None:
                # placement-probed block of the caller's workspace (mrphy_amd.workspace.GradWorkspace)
                hist_elems = max(int(lib.mrphy_blochsim_hist_bytes(code, N, nM, nT)), 16) // Mi.element_size()
                hist = ws.take_hist(hist_elems, dtype, device, (N, nM, nT))
                if not isinstance(hist, _hist.Hist):
                    hist = _hist.Hist([hist])
            else:
                hist = _hist.allocate(code, N, nM, nT, dtype, device)       # the allocator's: in parts (round 6, _hist.py)

        nul = _host.NULL_BC
        with torch.cuda.device(device):
            rc = lib.mrphy_blochsim_fwd_parts(
                code, Mi_c.data_ptr(), Beff_c.data_ptr(), *g.args,
                *(e1.args if e1 else nul), *(e2.args if e2 else nul),
                e1m1.t.data_ptr() if e1m1 else None,
                Mo.data_ptr(), *(hist.c_args() if need_hist else (None, 0, 0)),
                N, nM, nT, _host.current_stream(device))
        _lib.check(rc, 'mrphy_blochsim_fwd_parts')

        if need_hist:
            ctx.save_for_backward(Beff_c, g.t, *(x.t for x in (e1, e2) if x), *hist.tensors())
            ctx.hist_layout, ctx.n_consts = hist.layout, 1 + 2 * bool(e1)
            ctx.meta = (code, (g.sn, g.sm), (e1.sn, e1.sm) if e1 else None,
                        (e2.sn, e2.sm) if e2 else None, N, nM, nT, Beff.dtype)
            # shapes / dtypes for the constants' gradients (meta tensors: nothing of the caller's is kept
            # alive or hidden from autograd's version check)
            ctx.consts = tuple(None if c is None else torch.empty(c.shape, dtype=c.dtype, device='meta')
                               for c in (γ2πdt, E1, E2, E1_1))
            ctx.Nd = Nd
            ctx.ws, ctx.ws_gen = ws, (None if ws is None else ws.generation)
            ctx.relax = (e1, e2)          # for the domain check of the precise adjoint, made where it matters: in backward
        return Mo

    @staticmethod
    def backward(ctx, grad_Mo: Tensor):
        need_Mi, need_B = ctx.needs_input_grad[0:2]
        if not (need_Mi or need_B or any(ctx.needs_input_grad[2:6])):          # sims.py:156-157
            return None, None, None, None, None, None, None, None
        lib = _lib.require_library()
        saved = ctx.saved_tensors
        Beff_c, gt = saved[0], saved[1]
        hist = _hist.Hist(saved[1 + ctx.n_consts:], ctx.hist_layout)
        code, gs, e1s, e2s, N, nM, nT, beff_dtype = ctx.meta
        # E1, E2 != 0 for the precise adjoint (it divides by them once, the reference at every step, sims.py:174-177): checked
        # here, not in the forward -- which succeeds as the reference's does; one device read per constant set
        _host.require_invertible_relaxation(code, *ctx.relax, 'sims.blochsim')
        e1t, e2t = (saved[2], saved[3]) if e1s else (None, None)
        device, dtype = hist.device, hist.dtype

        need_c = ctx.needs_input_grad[2:6]
        gMo = grad_Mo.to(dtype).contiguous()
        gMi = torch.empty_like(gMo) if need_Mi else None
        ws = ctx.ws
        if ws is not None:                # the history must still be this forward's; grad_Beff is the workspace's block
            gB = ws.take_grad(Beff_c.shape, dtype, ctx.ws_gen)
            gB = gB if need_B else None
        else:
            gB = torch.empty_like(Beff_c) if need_B else None
        nul = _host.NULL_BC
        gC = torch.zeros((N, nM, 4), dtype=dtype, device=device) if any(need_c) else None
        gcs = (None, None, None, None)
        with torch.cuda.device(device):
            rc = lib.mrphy_blochsim_bwd_parts(
                code, *hist.c_args(), Beff_c.data_ptr(), gt.data_ptr(), *gs,
                *((e1t.data_ptr(),) + e1s if e1s else nul),
                *((e2t.data_ptr(),) + e2s if e2s else nul),
                gMo.data_ptr(), gMi.data_ptr() if need_Mi else None,
                gB.data_ptr() if need_B else None, gC.data_ptr() if gC is not None else None,
                N, nM, nT, _host.current_stream(device))
            _lib.check(rc, 'mrphy_blochsim_bwd_parts')
            if gC is not None:
                full = gC.reshape((N,) + tuple(ctx.Nd) + (4,))
                gcs = tuple(_reduce_to_const(full[..., i], c, N, ctx.Nd) if (want and c is not None) else None
                            for i, (c, want) in enumerate(zip(ctx.consts, need_c)))
        if need_B and gB.dtype != beff_dtype:
            gB = gB.to(beff_dtype)
        return (gMi, gB) + gcs + (None, None)


def _wants_grad(*xs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(x, Tensor) and x.requires_grad for x in xs)


@_host.half_via_float
def blochsim_consts(
    Mi: Tensor, Beff: Tensor, *,
    γ2πdt: Tensor, E1: Optional[Tensor] = None, E1_1: Optional[Tensor] = None,
    E2: Optional[Tensor] = None, workspace=None
) -> Tensor:
    r""":func:`blochsim` with the per-spin constants supplied by the caller, the way the
    reference's ``slowsims.blochsim_1step`` takes them (``slowsims.py:15-23``):
    ``γ2πdt = 2π·γ·dt``, ``E1 = exp(-dt/T1)``, ``E1_1 = E1 - 1``, ``E2 = exp(-dt/T2)``, each
    `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`; ``E1 = E1_1 = E2 = None`` disables relaxation.

    For callers that already hold the constants (repeated pulses on the same spins) and for
    comparing against results whose constants came from another ``exp()`` implementation.
    ``workspace``: as in :func:`blochsim`.
    """
    assert (Mi.shape[:-1] == Beff.shape[:-2])
    assert ((E1 is None) == (E2 is None) == (E1_1 is None))
    _host.require_device_tensor(Mi, 'Mi')
    Beff = Beff.to(Mi.device)
    _host.require_device_tensor(Beff, 'Beff')
    from . import workspace as _workspace
    want = _wants_grad(Mi, Beff, γ2πdt, E1, E2, E1_1)
    if workspace is None and want:
        workspace = _workspace.active(Beff.shape, Mi.dtype, Mi.device)
    return BlochSimHIP.apply(Mi, Beff, γ2πdt, E1, E2, E1_1, want, workspace)


@_host.half_via_float
def blochsim(
    Mi: Tensor, Beff: Tensor, *,
    T1: Optional[Tensor] = None, T2: Optional[Tensor] = None,
    γ: Tensor = γH, dt: Tensor = dt0, workspace=None
) -> Tensor:
    r"""Bloch simulator with explicit Jacobian operation, on the MI355X.

    Same contract as ``mrphy.sims.blochsim`` (``sims.py:272-315``):

    Usage:
        ``Mo = blochsim(Mi, Beff, *, T1, T2, γ, dt)``
        ``Mo = blochsim(Mi, Beff, *, T1=None, T2=None, γ, dt)``
    Inputs:
        - ``Mi``: `(N, *Nd, xyz)`, spins.
        - ``Beff``: `(N, *Nd, nT, xyz)`, "Gauss", B-effective; or the
          :class:`~mrphy_amd.beffective.LazyBeff` handle ``rfgr2beff(..., lazy=True)``
          returns, in which case the fused kernel runs and no ``Beff`` tensor ever exists.
    Optionals:
        - ``T1``, ``T2``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, "Sec"; both ``None`` = no relaxation.
        - ``γ``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, "Hz/Gauss" (default ``γH``).
        - ``dt``: `()` ⊻ `(N ⊻ 1,)`, "Sec" (default ``dt0``).
        - ``workspace``: a :class:`mrphy_amd.workspace.GradWorkspace` (extension): when a gradient is wanted, the
          history and ``grad_Beff`` are its placement-probed blocks instead of fresh allocations -- same bits, one
          forward / backward pair in flight per workspace.  Default: the workspace of an enclosing ``with ws:``.
    Outputs:
        - ``Mo``: `(N, *Nd, xyz)`.
    """
    from .beffective import LazyBeff
    from . import workspace as _workspace

    assert (Mi.shape[:-1] == Beff.shape[:-2])
    assert ((T1 is None) == (T2 is None))  # both or neither
    _host.require_device_tensor(Mi, 'Mi')
    if isinstance(Beff, LazyBeff):
        return Beff.blochsim(Mi, T1=T1, T2=T2, γ=γ, dt=dt)

    Beff = Beff.to(Mi.device)
    _host.require_device_tensor(Beff, 'Beff')
    # {γ, dt, T1, T2} -> rank of Beff by trailing singleton dims (sims.py:309-313), then the
    # constants with the reference's own expressions (sims.py:62,74-76), on the tensors' device
    γ2πdt, E1, E2, E1_1 = relax_constants(T1, T2, γ, dt, Beff.ndim, Mi.device)
    if workspace is None and _wants_grad(Mi, Beff):
        workspace = _workspace.active(Beff.shape, Mi.dtype, Mi.device)
    return BlochSimHIP.apply(Mi, Beff, γ2πdt, E1, E2, E1_1, _wants_grad(Mi, Beff), workspace)


class FreePrecHIP(Function):
    r"""``Mo = FreePrecHIP.apply(Mi, dur, T1, T2, Δf)`` -- see :func:`freeprec`."""

    @staticmethod
    def _launch(fn, Min, dur, T1, T2, Δf):
        lib = _lib.require_library()
        device, dtype = Min.device, Min.dtype
        N, Nd = Min.shape[0], tuple(Min.shape[1:-1])
        nM = prod(Nd)
        x = Min.detach().contiguous()
        out = torch.empty_like(x)
        d = dur.detach().to(device=device, dtype=dtype).reshape(-1).contiguous()
        assert d.numel() in (1, N), "dur must be () or (N ⊻ 1,)"
        mk = lambda c: None if c is None else _host.Bcast(c.detach(), N, Nd, dtype, device)  # noqa
        t1, t2, df = mk(T1), mk(T2), mk(Δf)
        nul = _host.NULL_BC
        code = _lib.F64 if dtype == torch.float64 else _lib.F32
        with torch.cuda.device(device):
            rc = getattr(lib, fn)(code, x.data_ptr(), d.data_ptr(), 1 if (d.numel() == N and N > 1) else 0,
                                  *(t1.args if t1 else nul), *(t2.args if t2 else nul),
                                  *(df.args if df else nul), out.data_ptr(), N, nM,
                                  _host.current_stream(device))
        _lib.check(rc, fn)
        return out

    @staticmethod
    def forward(ctx, Mi, dur, T1, T2, Δf, consts_grad=False):
        ctx.args = (dur, T1, T2, Δf)
        ctx.consts_grad = bool(consts_grad)
        if ctx.consts_grad:                               # slowsims.freeprec: dL/d{dur, T1, T2, Δf} too
            ctx.save_for_backward(Mi.detach())
        return FreePrecHIP._launch('mrphy_freeprec_fwd', Mi, dur, T1, T2, Δf)

    @staticmethod
    def backward(ctx, grad_Mo):
        need = ctx.needs_input_grad
        gM = FreePrecHIP._launch('mrphy_freeprec_bwd', grad_Mo, *ctx.args) if need[0] else None   # sims.py:397-398
        gcs = (None,) * 4
        if ctx.consts_grad and any(need[1:5]):
            lib = _lib.require_library()
            (Mi,) = ctx.saved_tensors
            dur, T1, T2, Δf = ctx.args
            device, dtype = Mi.device, Mi.dtype
            N, Nd = Mi.shape[0], tuple(Mi.shape[1:-1])
            nM = prod(Nd)
            x, g = Mi.contiguous(), grad_Mo.detach().to(dtype).contiguous()
            d = dur.detach().to(device=device, dtype=dtype).reshape(-1).contiguous()
            mk = lambda c: None if c is None else _host.Bcast(c.detach(), N, Nd, dtype, device)  # noqa
            t1, t2, df = mk(T1), mk(T2), mk(Δf)
            nul = _host.NULL_BC
            gC = torch.empty((N * nM, 4), dtype=dtype, device=device)
            with torch.cuda.device(device):
                rc = lib.mrphy_freeprec_bwd_consts(
                    _lib.F64 if dtype == torch.float64 else _lib.F32, x.data_ptr(), g.data_ptr(), d.data_ptr(),
                    1 if (d.numel() == N and N > 1) else 0, *(t1.args if t1 else nul), *(t2.args if t2 else nul),
                    *(df.args if df else nul), gC.data_ptr(), N, nM, _host.current_stream(device))
            _lib.check(rc, 'mrphy_freeprec_bwd_consts')
            full = gC.reshape((N,) + Nd + (4,))
            gcs = tuple(_reduce_to_const(full[..., i], c, N, Nd) if (want and c is not None) else None
                        for i, (c, want) in enumerate(zip((dur, T1, T2, Δf), need[1:5])))
        return (gM,) + gcs + (None,)


@_host.half_via_float
def freeprec(
    Mi: Tensor, dur: Tensor, *,
    T1: Optional[Tensor] = None, T2: Optional[Tensor] = None,
    Δf: Optional[Tensor] = None
) -> Tensor:
    r"""Isochromats free precession with given relaxation and off-resonance, on the MI355X.

    Same contract as ``mrphy.sims.freeprec`` (``sims.py:424-458``); differentiable w.r.t.
    ``Mi`` only, like the reference (``sims.py:321``).

    Usage:
        ``Mo = freeprec(Mi, dur, *, T1, T2, Δf)``
    Inputs:
        - ``Mi``: `(N, *Nd, xyz)`, spins.
        - ``dur``: `()` ⊻ `(N ⊻ 1,)`, "Sec", duration of free precession.
    Optionals:
        - ``T1``, ``T2``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, "Sec"; both ``None`` = no relaxation.
        - ``Δf``: `(N ⊻ 1, *Nd ⊻ 1,)`, "Hz", off-resonance; ``None`` = no precession.
    Outputs:
        - ``Mo``: `(N, *Nd, xyz)`.
    """
    assert ((T1 is None) == (T2 is None))  # both or neither
    _host.require_device_tensor(Mi, 'Mi')
    return FreePrecHIP.apply(Mi, dur, T1, T2, Δf)
