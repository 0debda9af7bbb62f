r"""Fused ``rf, gr -> Mo``: ``rfgr2beff`` + ``blochsim`` in one kernel, no ``Beff`` in HBM.

This is what ``mrphy.mobjs.SpinArray.applypulse`` does back-to-back (reference
``mobjs.py:435-446``: ``pulse2beff`` then ``sims.blochsim``).  The per-step field is assembled
in registers from the wave-uniform pulse sample and the lane's own ``loc, Δf/γ, b1Map``
(``beffective.py:137-165``), then the same step as ``mrphy_blochsim_fwd`` is applied; the result is
bit-identical to the two-kernel path.

Gradients w.r.t. ``Mi``, ``rf`` and ``gr`` (what pulse design differentiates) use the fused
adjoint ``mrphy_blochsim_rfgr_bwd``: the forward leaves a checkpoint of ``M`` every 16 steps
(0.75 B per spin-step instead of the 12 B/spin-step history plus the 24 B/spin-step of ``Beff`` and
``grad_Beff`` of the two-kernel path), each segment is recomputed in registers and swept backwards,
and ``grad_rf``/``grad_gr`` come out of a deterministic reduction over spins; parallel transmit
(``rf`` `(N,xy,nT,nCoils)` with a ``b1Map``, up to 8 coils) has its own kernel.  Cases the fused
adjoint does not cover (more coils, gradients w.r.t. the spin-side maps) compose ``rfgr2beff`` and
``blochsim`` instead -- HIP kernels as well; a pulse length that is not a multiple of 16 is split
into a fused part and a tail of at most 15 composed steps.
"""
from math import pi as π, prod  # noqa: F401
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host
from ._consts import γH, dt0

__all__ = ['blochsim_rfgr']


class BlochSimRfGrHIP(Function):
    r"""``Mo = BlochSimRfGrHIP.apply(Mi, rf, gr, pulse_on_spins, γ2πdt, E1, E2, E1_1, want_ckpt)``

    ``want_ckpt`` is decided by the caller (:func:`blochsim_rfgr`: a gradient w.r.t. ``Mi``, ``rf``
    or ``gr`` is wanted, grad mode is on and the fused adjoint covers the case) -- NOT re-derived
    from ``ctx.needs_input_grad``, which stays ``True`` under ``torch.no_grad()``."""

    @staticmethod
    def forward(ctx, Mi, rf, gr, p, γ2πdt, E1, E2, E1_1, want_ckpt=False):
        from . import sims
        lib = _lib.require_library()
        device, dtype = Mi.device, Mi.dtype
        code, g, e1, e2, e1m1 = sims._prep_constants(γ2πdt, E1, E2, E1_1, p.N, p.Nd, dtype, device)
        Mi_c = Mi.detach().contiguous()
        Mo = torch.empty_like(Mi_c)
        need = bool(want_ckpt)
        ck = int(lib.mrphy_blochsim_rfgr_ck_every())
        # one checkpoint per started segment: nCk = ceil(nT / ck_every) (include/mrphy_hip.h)
        Mck = (torch.empty((-(-p.nT // ck), p.N * p.nM, 3), dtype=dtype, device=device)
               if need else None)
        nul = _host.NULL_BC
        consts = (*g.args, *(e1.args if e1 else nul), *(e2.args if e2 else nul),
                  e1m1.t.data_ptr() if e1m1 else None)
        with torch.cuda.device(device):
            rc = lib.mrphy_blochsim_rfgr_fwd(
                code, Mi_c.data_ptr(), *p.k0_args(), *consts, Mo.data_ptr(),
                Mck.data_ptr() if need else None, ck if need else 0,
                p.N, p.nM, p.nT, p.nC, _host.current_stream(device))
        _lib.check(rc, 'mrphy_blochsim_rfgr_fwd')
        if need:
            ctx.save_for_backward(Mck)
            ctx.keep = (p, code, consts, (g, e1, e2, e1m1), rf.shape, gr.shape, rf.dtype, gr.dtype)
        return Mo

    @staticmethod
    def backward(ctx, grad_Mo):
        from .beffective import _fold_pulse_grad
        need_Mi, need_rf, need_gr = ctx.needs_input_grad[0:3]
        if not (need_Mi or need_rf or need_gr):
            return (None,) * 9
        lib = _lib.require_library()
        (Mck,) = ctx.saved_tensors
        p, code, consts, _alive, rf_shape, gr_shape, rf_dtype, gr_dtype = ctx.keep
        # the precise adjoint divides by E once (as the reference's does at every step): refuse E == 0 HERE, where it
        # matters -- the forward succeeds as the reference's does; cached per constants
        _host.require_invertible_relaxation(code, _alive[1], _alive[2], 'fused.blochsim_rfgr')
        device, dtype = Mck.device, Mck.dtype
        gMo = grad_Mo.to(dtype).contiguous()
        gMi = torch.empty_like(gMo) if need_Mi else None
        g_rf = torch.empty((p.N, 2, p.nT, p.nC), dtype=dtype, device=device) if need_rf else None
        g_gr = torch.empty((p.N, 3, p.nT), dtype=dtype, device=device) if need_gr else None
        outs = (gMi.data_ptr() if need_Mi else None, g_rf.data_ptr() if need_rf else None,
                g_gr.data_ptr() if need_gr else None)
        if p.nC == 1:
            nbytes = int(lib.mrphy_blochsim_rfgr_bwd_workspace(code, p.N, p.nM, p.nT))
            work = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                rc = lib.mrphy_blochsim_rfgr_bwd(
                    code, Mck.data_ptr(), *p.k0_args(), *consts, gMo.data_ptr(), *outs,
                    work.data_ptr(), work.numel(), p.N, p.nM, p.nT, _host.current_stream(device))
            _lib.check(rc, 'mrphy_blochsim_rfgr_bwd')
        else:                                   # parallel transmit: per-coil sums in the kernel
            nbytes = int(lib.mrphy_blochsim_rfgr_mc_bwd_workspace(code, p.N, p.nM, p.nT, p.nC))
            work = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                rc = lib.mrphy_blochsim_rfgr_mc_bwd(
                    code, Mck.data_ptr(), *p.k0_args(), *consts, gMo.data_ptr(), *outs,
                    work.data_ptr(), work.numel(), p.N, p.nM, p.nT, p.nC,
                    _host.current_stream(device))
            _lib.check(rc, 'mrphy_blochsim_rfgr_mc_bwd')
        return (gMi,
                _fold_pulse_grad(g_rf, rf_shape, rf_dtype, p.b1 is None) if need_rf else None,
                _fold_pulse_grad(g_gr, gr_shape, gr_dtype, False) if need_gr else None,
                None, None, None, None, None, None)


@_host.half_via_float
def blochsim_rfgr(
    Mi: Tensor, rf: Tensor, gr: Tensor, loc: Tensor, *,
    Δf: Optional[Tensor] = None, b1Map: Optional[Tensor] = None, γ_beff: Tensor = γH,
    T1: Optional[Tensor] = None, T2: Optional[Tensor] = None,
    γ: Tensor = γH, dt: Tensor = dt0, consts: Optional[dict] = None
) -> Tensor:
    r"""``blochsim(Mi, rfgr2beff(rf, gr, loc, Δf=Δf, b1Map=b1Map, γ=γ_beff), T1=T1, T2=T2,
    γ=γ, dt=dt)`` without the intermediate tensor, differentiable w.r.t. ``Mi``, ``rf``, ``gr``.

    ``Mi``: `(N, *Nd, xyz)`; the other arguments as in
    :func:`mrphy_amd.beffective.rfgr2beff` and :func:`mrphy_amd.sims.blochsim`.  ``consts``
    (``dict(γ2πdt=, E1=, E1_1=, E2=)``) replaces ``T1, T2, γ, dt`` as in
    :func:`mrphy_amd.sims.blochsim_consts`.
    """
    from . import beffective, sims
    _host.require_device_tensor(Mi, 'Mi')
    assert (T1 is None) == (T2 is None)
    lib = _lib.require_library()
    grad_on = torch.is_grad_enabled()
    rq = lambda x: grad_on and isinstance(x, Tensor) and x.requires_grad  # noqa: E731
    maps_grad = any(rq(x) for x in (loc, Δf, b1Map))
    pulse_grad = any(rq(x) for x in (Mi, rf, gr))
    p = beffective._PulseOnSpins(rf.detach(), gr.detach(), loc.detach(),
                                 None if Δf is None else Δf.detach(),
                                 None if b1Map is None else b1Map.detach(), γ_beff.detach())
    seg = int(lib.mrphy_blochsim_rfgr_ck_every())
    seg_ok = p.nT % seg == 0
    one_coil = p.nC == 1 and (rf.ndim == 3 or b1Map is not None or rf.shape[-1] == 1)
    ptx = 1 < p.nC <= int(lib.mrphy_blochsim_rfgr_mc_max_coils()) and p.b1 is not None
    fused_adjoint_ok = seg_ok and (one_coil or ptx)
    if pulse_grad and not maps_grad and not seg_ok and (one_coil or ptx) and p.nT > seg:
        # A pulse length that is not a whole number of checkpoint segments: the first
        # floor(nT/16)*16 steps go through the fused pair, the remaining <= 15 steps through
        # rfgr2beff + blochsim, whose Beff is then tiny.  Autograd chains the two; the forward is
        # the same step arithmetic throughout (bit-identical to a single pass).
        n1 = (p.nT // seg) * seg
        kw = dict(Δf=Δf, b1Map=b1Map, γ_beff=γ_beff, T1=T1, T2=T2, γ=γ, dt=dt, consts=consts)
        M1 = blochsim_rfgr(Mi, rf[:, :, :n1], gr[:, :, :n1], loc, **kw)
        return blochsim_rfgr(M1, rf[:, :, n1:], gr[:, :, n1:], loc, **kw)
    # fp64 with more than 8 transmit coils: the fused forward has no register build for it (it would spill),
    # the composed route does (k_rfgr2beff_steps / _pk + K1) and gives the same bits
    wide_f64 = (p.dtype == torch.float64 and p.nC > 8) or (p.b1 is not None and p.nC > 64)   # (> 64: coil-blocked K0 + K1)
    if maps_grad or wide_f64 or (pulse_grad and not fused_adjoint_ok):
        beff = beffective.rfgr2beff(rf, gr, loc, Δf=Δf, b1Map=b1Map, γ=γ_beff, lazy=False)
        if consts is not None:
            return sims.blochsim_consts(Mi, beff, **consts)
        return sims.blochsim(Mi, beff, T1=T1, T2=T2, γ=γ, dt=dt)

    device, dtype = Mi.device, Mi.dtype
    assert p.device == device and p.dtype == dtype, "Mi and loc must share device and dtype"
    assert tuple(Mi.shape[:-1]) == (p.N,) + p.Nd
    if consts is not None:
        γ2πdt, E1, E2, E1_1 = (consts.get(k) for k in ('γ2πdt', 'E1', 'E2', 'E1_1'))
    else:
        γ2πdt, E1, E2, E1_1 = sims.relax_constants(T1, T2, γ, dt, 1 + len(p.Nd) + 2, device)
    return BlochSimRfGrHIP.apply(Mi, rf, gr, p, γ2πdt, E1, E2, E1_1,
                                 pulse_grad and fused_adjoint_ok)
