r"""B-effective assembly, HIP-backed.

Drop-in for ``mrphy.beffective.rfgr2beff`` (reference ``mrphy/beffective.py:107-168``) and
``mrphy.beffective.beff2uϕ`` (``beffective.py:18-37``).

``rfgr2beff`` writes the ``(N, *Nd, nT, xyz)`` tensor with one HBM-write-bound kernel
(``mrphy_rfgr2beff``); it is differentiable (the reference relies on autograd through plain
ops: ``loc @ gr``, the complex b1Map product, ``stack``): ``rf`` and ``gr`` gradients come
from a deterministic HIP reduction over spins (``mrphy_rfgr2beff_bwd``).

With ``lazy=True`` nothing is written: a :class:`LazyBeff` handle is returned that
``mrphy_amd.sims.blochsim`` consumes with the fused ``rf,gr -> Mo`` kernel.  At 128^3 spins x
4096 steps the materialised tensor is 103 GB; the handle is a few pointers.
"""
from math import pi as π, prod  # noqa: F401
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _host
from ._consts import γH, dt0

__all__ = ['rfgr2beff', 'beff2uϕ', 'beff2ab', 'LazyBeff']

LAZY_DEFAULT = False     # set by mrphy_amd.install(lazy_beff=True)


# ---------------------------------------------------------------------------------------------
# argument normalisation shared by the eager kernel and the lazy handle
# ---------------------------------------------------------------------------------------------
class _PulseOnSpins:
    r"""rf, gr, loc, Δf, b1Map, γ normalised to the C-ABI layouts (see ``mrphy_hip.h`` K0)."""

    def __init__(self, rf, gr, loc, Δf, b1Map, γ):
        assert (rf.device == gr.device == loc.device)      # beffective.py:131
        _host.require_device_tensor(loc, 'loc')
        _host.require_device_tensor(rf, 'rf')
        _host.require_device_tensor(gr, 'gr')
        device, dtype = loc.device, loc.dtype
        shape = tuple(loc.shape)
        N, Nd = shape[0], shape[1:-1]
        assert shape[-1] == 3 and gr.shape[1] == 3 and rf.shape[1] == 2
        nT = gr.shape[2]
        assert rf.shape[2] == nT
        assert rf.shape[0] in (1, N) and gr.shape[0] in (1, N)

        rf, gr = rf.to(dtype), gr.to(dtype)
        if b1Map is None:
            if rf.ndim == 4:                 # multi-coil rf, no map: coils add (beffective.py:148)
                rf = rf.sum(dim=-1)
            rf4 = rf[..., None]
            b1 = None
        else:
            b1 = b1Map.to(device=device, dtype=dtype)
            if b1.ndim == 1 + len(Nd) + 1:   # (N, *Nd, xy) -> (N, *Nd, xy, 1)
                b1 = b1[..., None]
            rf4 = rf if rf.ndim == 4 else rf[..., None]
            assert b1.shape[-1] == rf4.shape[-1], "b1Map and rf disagree on nCoils"
            # the map may broadcast over batch and spins (tests/test_sims.py:52 passes (N,1,2,1))
            assert b1.shape[-2] == 2 and b1.shape[0] in (1, N) and \
                all(a in (1, d) for a, d in zip(b1.shape[1:-2], Nd))
            b1 = b1.expand((N,) + tuple(Nd) + tuple(b1.shape[-2:]))
        self.N, self.Nd, self.nM, self.nT, self.nC = N, tuple(Nd), prod(Nd), nT, rf4.shape[-1]
        self.device, self.dtype = device, dtype
        self.rf, self.gr = rf4.contiguous(), gr.contiguous()
        self.loc = loc.contiguous()
        self.b1 = None if b1 is None else b1.contiguous()
        self.rf_sn = self.rf.stride(0) if (self.rf.shape[0] == N and N > 1) else 0
        self.gr_sn = self.gr.stride(0) if (self.gr.shape[0] == N and N > 1) else 0
        if Δf is not None:
            Δf = Δf.to(device)
            assert Δf.ndim <= 1 + len(Nd), "Δf must be (N ⊻ 1, *Nd ⊻ 1)"
            self.df = _host.Bcast(Δf, N, self.Nd, dtype, device)
            self.gam = _host.Bcast(γ.to(device), N, self.Nd, dtype, device)
            self.df_v = _host.pad_trailing(Δf.to(dtype), 1 + len(Nd))
            self.gam_v = _host.pad_trailing(γ.to(device=device, dtype=dtype), 1 + len(Nd))
        else:
            self.df = self.gam = None

    def k0_args(self):
        nul = _host.NULL_BC
        return (self.rf.data_ptr(), self.rf_sn, self.gr.data_ptr(), self.gr_sn,
                self.loc.data_ptr(),
                *(self.df.args if self.df else nul), *(self.gam.args if self.gam else nul),
                self.b1.data_ptr() if self.b1 is not None else None)


def _code(dtype):
    return _lib.F64 if dtype == torch.float64 else _lib.F32


# cache policy of K0's stores (include/mrphy_hip.h: MRPHY_STORE_*)
STORE_POLICIES = {None: -1, 'auto': -1, 'plain': 0, 'nt': 1, 'sc1nt': 2}


class _Boxed:
    r"""A tensor handed to an autograd Function WITHOUT becoming one of its inputs."""
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t


class RfGr2BeffHIP(Function):
    r"""``beff = RfGr2BeffHIP.apply(rf, gr, loc, Δf, b1Map, γ[, out[, store]])``"""

    @staticmethod
    def forward(ctx, rf, gr, loc, Δf, b1Map, γ, out=None, store=None):
        assert store in STORE_POLICIES, f"rfgr2beff: store must be one of {sorted(map(str, STORE_POLICIES))}"
        lib = _lib.require_library()
        p = _PulseOnSpins(rf.detach(), gr.detach(), loc.detach(),
                          None if Δf is None else Δf.detach(),
                          None if b1Map is None else b1Map.detach(), γ.detach())
        shape = (p.N,) + p.Nd + (p.nT, 3)
        if out is None:
            beff = torch.empty(shape, dtype=p.dtype, device=p.device)
        else:                                      # the caller's block (mrphy_amd.workspace.BeffArena / GradWorkspace)
            out = out.t
            assert tuple(out.shape) == shape and out.dtype == p.dtype and out.device == p.device and \
                out.is_contiguous() and not out.requires_grad, \
                f"rfgr2beff: out must be a contiguous {shape} {p.dtype} tensor on {p.device}"
            # A NEW tensor over the block's memory, not the caller's tensor object: the block is not an input of
            # this autograd node (it arrives boxed), so a loop that differentiates through the same block iteration
            # after iteration starts a fresh graph each time (round 4 passed the tensor itself and marked it dirty:
            # the second differentiable call found its `out` carrying the first call's graph).
            beff = out.detach()
        with torch.cuda.device(p.device):
            rc = lib.mrphy_rfgr2beff_st(_code(p.dtype), *p.k0_args(), beff.data_ptr(),
                                        p.N, p.nM, p.nT, p.nC, STORE_POLICIES[store], _host.current_stream(p.device))
        _lib.check(rc, 'mrphy_rfgr2beff_st')
        if out is not None:
            # the block was rewritten through a raw pointer: say so to autograd.  `beff`, `out` and every
            # `detach()` of them share one version counter, so a graph that SAVED the block's previous contents (an
            # earlier sims.blochsim on the same block whose backward has not run yet) now raises "modified by an inplace
            # operation" in its backward instead of silently differentiating the wrong field.
            torch.autograd.graph.increment_version(out)
        ctx.p = p
        ctx.in_shapes = (rf.shape, gr.shape, rf.dtype, gr.dtype)
        ctx.had = (Δf is not None, b1Map is not None)
        ctx.orig = (None if Δf is None else Δf.shape, None if b1Map is None else b1Map.shape,
                    γ.shape)
        return beff

    @staticmethod
    def backward(ctx, gB):
        need = ctx.needs_input_grad
        p = ctx.p
        g_rf = g_gr = g_loc = g_df = g_b1 = g_γ = None
        gB = gB.to(p.dtype).contiguous()
        if need[0] or need[1]:
            g_rf, g_gr = _rfgr_grads(p, gB, need[0], need[1], ctx.in_shapes)
        # Rarely needed gradients w.r.t. the spin-side maps: plain torch ops on the device
        # (off the pulse-design hot path, which differentiates rf/gr only).
        full = (p.N,) + p.Nd
        gB3 = gB.reshape(p.N, p.nM, p.nT, 3)
        if need[2]:
            g_loc = torch.einsum('nst,nit->nsi', gB3[..., 2], p.gr.expand(p.N, 3, p.nT)) \
                .reshape(full + (3,))
        if ctx.had[0] and (need[3] or need[5]):
            gz = gB3[..., 2].sum(dim=-1).reshape(full)
            gam, df = p.gam_v, p.df_v           # views padded to rank 1+len(Nd)
            if need[3]:
                g_df = _sum_to(gz / gam, ctx.orig[0])
            if need[5]:
                g_γ = _sum_to(-gz * df / (gam * gam), ctx.orig[2])
        if ctx.had[1] and need[4]:
            rfe = p.rf.expand(p.N, 2, p.nT, p.nC)
            gx, gy = gB3[..., 0], gB3[..., 1]
            g_b1r = torch.einsum('nst,ntc->nsc', gx, rfe[:, 0]) + \
                torch.einsum('nst,ntc->nsc', gy, rfe[:, 1])
            g_b1i = torch.einsum('nst,ntc->nsc', gy, rfe[:, 0]) - \
                torch.einsum('nst,ntc->nsc', gx, rfe[:, 1])
            g_b1 = torch.stack([g_b1r, g_b1i], dim=-2).reshape(full + (2, p.nC))
            g_b1 = _sum_to(g_b1, ctx.orig[1])
        return g_rf, g_gr, g_loc, g_df, g_b1, g_γ, None, None


def _sum_to(x: Tensor, shape) -> Tensor:
    r"""Reduce a broadcast gradient back to the shape of the operand that was broadcast
    (operands are right-padded with singleton dims, as the reference pads them)."""
    shape, full = tuple(shape), tuple(x.shape)
    padded = shape + (1,) * (len(full) - len(shape))
    dims = [i for i, (a, b) in enumerate(zip(full, padded)) if b == 1 and a != 1]
    if dims:
        x = x.sum(dim=dims, keepdim=True)
    return x.reshape(shape)


def _rfgr_grads(p: _PulseOnSpins, gB: Tensor, need_rf: bool, need_gr: bool, in_shapes):
    r"""grad_rf, grad_gr from ``grad_beff`` with the deterministic HIP reduction."""
    lib = _lib.require_library()
    rf_shape, gr_shape, rf_dtype, gr_dtype = in_shapes
    code = _code(p.dtype)
    nbytes = lib.mrphy_rfgr2beff_bwd_workspace(code, p.N, p.nM, p.nT, p.nC)
    work = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=p.device)
    g_rf = torch.empty((p.N, 2, p.nT, p.nC), dtype=p.dtype, device=p.device) if need_rf else None
    g_gr = torch.empty((p.N, 3, p.nT), dtype=p.dtype, device=p.device) if need_gr else None
    with torch.cuda.device(p.device):
        rc = lib.mrphy_rfgr2beff_bwd(
            code, gB.data_ptr(), p.loc.data_ptr(),
            p.b1.data_ptr() if p.b1 is not None else None,
            g_rf.data_ptr() if need_rf else None, g_gr.data_ptr() if need_gr else None,
            work.data_ptr(), work.numel(), p.N, p.nM, p.nT, p.nC,
            _host.current_stream(p.device))
    _lib.check(rc, 'mrphy_rfgr2beff_bwd')
    return (_fold_pulse_grad(g_rf, rf_shape, rf_dtype, p.b1 is None) if need_rf else None,
            _fold_pulse_grad(g_gr, gr_shape, gr_dtype, False) if need_gr else None)


def _fold_pulse_grad(g: Tensor, shape, dtype, coil_broadcast: bool) -> Tensor:
    r"""Per-batch gradient (N, k, nT[, nC]) -> the caller's rf/gr shape: a batch-1 pulse that
    was broadcast over N spins sums over N; a coil axis that was summed away (multi-coil rf
    without b1Map) gets the same gradient on every coil; a missing coil axis is dropped."""
    shape = tuple(shape)
    if g.shape[0] != shape[0]:
        g = g.sum(dim=0, keepdim=True)
    if g.ndim == 4:
        if len(shape) == 3:
            g = g[..., 0]
        elif coil_broadcast and shape[3] != g.shape[3]:
            g = g.expand(g.shape[:3] + (shape[3],))
    return g.to(dtype).reshape(shape) if g.shape == shape else g.to(dtype).expand(shape).clone()


# ---------------------------------------------------------------------------------------------
class LazyBeff:
    r"""What ``rfgr2beff(..., lazy=True)`` returns: the *recipe* for ``beff``, not the tensor.

    It quacks like the ``(N, *Nd, nT, xyz)`` tensor as far as ``mrphy.mobjs`` and
    ``sims.blochsim`` look at it before simulating (``shape``, ``ndim``, ``device``,
    ``dtype``, ``.to(device)``: ``mobjs.py:653-654``, ``sims.py:305-306``); ``blochsim``
    recognises it and runs the fused kernel.  Anything else -- indexing, arithmetic, passing
    it to a torch function -- goes through :meth:`materialize` (the eager K0 kernel).
    """

    def __init__(self, rf, gr, loc, Δf, b1Map, γ):
        self._inputs = (rf, gr, loc, Δf, b1Map, γ)
        N, Nd = loc.shape[0], tuple(loc.shape[1:-1])
        self.shape = torch.Size((N,) + Nd + (gr.shape[2], 3))
        self.ndim = len(self.shape)
        self.device, self.dtype = loc.device, loc.dtype
        self._dense = None

    def dim(self):
        return self.ndim

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def to(self, *args, **kw):
        device = kw.get('device', args[0] if args and not isinstance(args[0], torch.dtype)
                        else None)
        dtype = kw.get('dtype', next((a for a in args if isinstance(a, torch.dtype)), None))
        same_dev = device is None or torch.device(device) == self.device or \
            (torch.device(device).type == self.device.type and torch.device(device).index is None)
        if same_dev and (dtype is None or dtype == self.dtype):
            return self
        return self.materialize().to(*args, **kw)

    def materialize(self) -> Tensor:
        if self._dense is None:
            rf, gr, loc, Δf, b1Map, γ = self._inputs
            self._dense = RfGr2BeffHIP.apply(rf, gr, loc, Δf, b1Map, γ)
        return self._dense

    def blochsim(self, Mi, *, T1=None, T2=None, γ=γH, dt=dt0):
        from .fused import blochsim_rfgr
        rf, gr, loc, Δf, b1Map, γ_b = self._inputs
        return blochsim_rfgr(Mi, rf, gr, loc, Δf=Δf, b1Map=b1Map, γ_beff=γ_b,
                             T1=T1, T2=T2, γ=γ, dt=dt)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        conv = lambda a: a.materialize() if isinstance(a, LazyBeff) else a  # noqa: E731
        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in kwargs.items()})

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __getattr__(self, name):           # any other tensor attribute/method
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __repr__(self):
        return f"LazyBeff(shape={tuple(self.shape)}, dtype={self.dtype}, device={self.device})"


@_host.half_via_float
def rfgr2beff(
    rf: Tensor,
    gr: Tensor,
    loc: Tensor, *,
    Δf: Optional[Tensor] = None,
    b1Map: Optional[Tensor] = None,
    γ: Tensor = γH,
    lazy: Optional[bool] = None,
    out: Optional[Tensor] = None,
    store: Optional[str] = None
):
    r"""Compute B-effectives from rf and gradients, on the MI355X.

    Same contract as ``mrphy.beffective.rfgr2beff`` (``beffective.py:107-168``):

    Usage:
        ``beff = rfgr2beff(rf, gr, loc, *, Δf, b1Map, γ)``
    Inputs:
        - ``rf``: `(N,xy,nT,(nCoils))`, "Gauss", ``xy`` separates real and imaginary part.
        - ``gr``: `(N,xyz,nT)`, "Gauss/cm".
        - ``loc``: `(N,*Nd,xyz)`, "cm", locations.
    Optionals:
        - ``Δf``: `(N,*Nd,)`, "Hz", off-resonance.
        - ``b1Map``: `(N,*Nd,xy,(nCoils))`, a.u., transmit sensitivity.  (The reference's
          b1Map branch only works for the compact 1-D ``Nd`` -- ``beffective.py:163-164`` --
          any ``Nd`` works here.)
        - ``γ``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, "Hz/Gauss", gyromagnetic ratio.
        - ``lazy``: return a :class:`LazyBeff` handle instead of the tensor (extension).
        - ``out``: write into this contiguous `(N,*Nd,nT,xyz)` tensor's memory and return a tensor over it
          (extension; what :class:`mrphy_amd.workspace.BeffArena` / ``GradWorkspace`` hand out).
        - ``store``: cache policy of the kernel's stores, ``None`` / ``'auto'`` (by size), ``'plain'``, ``'nt'``,
          ``'sc1nt'`` (extension; never changes the result -- ``BeffArena`` times the candidates and reports the faster).
    Outputs:
        - ``beff``: `(N,*Nd,nT,xyz)`, "Gauss".
    """
    assert (rf.device == gr.device == loc.device)
    _host.require_device_tensor(loc, 'loc')
    if LAZY_DEFAULT if lazy is None else lazy:
        assert out is None and store is None, "rfgr2beff: lazy=True writes nothing, out= / store= have no meaning"
        return LazyBeff(rf, gr, loc, Δf, b1Map, γ)
    return RfGr2BeffHIP.apply(rf, gr, loc, Δf, b1Map, γ, None if out is None else _Boxed(out), store)


class _Beff2UPhi(Function):
    r"""``U, Φ = _Beff2UPhi.apply(beff, γ2πdt)`` (xyz last) with the explicit adjoint
    ``mrphy_beff2uphi_bwd``; the reference differentiates ``F.normalize`` and ``torch.norm``
    (``beffective.py:35-36``)."""

    @staticmethod
    def forward(ctx, beff, γ2πdt):
        lib = _lib.require_library()
        b = beff.detach().contiguous()
        N, Nd = b.shape[0], tuple(b.shape[1:-1])
        nM = prod(Nd)
        cdt = torch.float64 if (γ2πdt.dtype == torch.float64 or b.dtype == torch.float64) \
            else torch.float32
        g = _host.Bcast(γ2πdt.detach().to(b.device), N, Nd, cdt, b.device)
        U = torch.empty_like(b)
        Φ = torch.empty(b.shape[:-1], dtype=b.dtype, device=b.device)
        code = _host.dtype_code(b.dtype, cdt)
        with torch.cuda.device(b.device):
            rc = lib.mrphy_beff2uphi(code, b.data_ptr(), *g.args, U.data_ptr(), Φ.data_ptr(), N, nM,
                                     _host.current_stream(b.device))
        _lib.check(rc, 'mrphy_beff2uphi')
        ctx.save_for_backward(b, g.t)
        ctx.meta = (code, (g.sn, g.sm), N, nM, beff.dtype, γ2πdt.shape, γ2πdt.dtype)
        return U, Φ

    @staticmethod
    def backward(ctx, gU, gΦ):
        lib = _lib.require_library()
        b, gt = ctx.saved_tensors
        code, gs, N, nM, b_dtype, γ_shape, γ_dtype = ctx.meta
        need_b, need_γ = ctx.needs_input_grad
        gUc = None if gU is None else gU.to(b.dtype).contiguous()
        gΦc = None if gΦ is None else gΦ.to(b.dtype).contiguous()
        gb = torch.empty_like(b) if need_b else None
        gg = torch.empty(b.shape[:-1], dtype=b.dtype, device=b.device) if need_γ else None
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(b.device):
            rc = lib.mrphy_beff2uphi_bwd(code, b.data_ptr(), gt.data_ptr(), *gs, ptr(gUc), ptr(gΦc),
                                         ptr(gb), ptr(gg), N, nM, _host.current_stream(b.device))
        _lib.check(rc, 'mrphy_beff2uphi_bwd')
        if need_b:
            gb = gb.to(b_dtype)
        if need_γ:                  # per-spin gradient -> γ2πdt's broadcast shape
            gg = _sum_to(gg, γ_shape).to(γ_dtype)
        return gb, gg


def beff2uϕ(beff: Tensor, γ2πdt: Tensor, *, dim=-1) -> Tuple[Tensor, Tensor]:
    r"""Rotation axes and angles from B-effectives (``beffective.py:18-37``).

    ``U = beff/max(‖beff‖, 1e-12)``, ``Φ = -‖beff‖·γ2πdt`` (negated: the Bloch equation is
    ``M×B``).  ``beff``: `(N, *Nd, xyz)` -> ``U``: `(N, *Nd, xyz)`, ``Φ``: `(N, *Nd)`.
    Differentiable w.r.t. ``beff`` and ``γ2πdt`` (explicit adjoint kernel).
    """
    _host.require_device_tensor(beff, 'beff')
    moved = dim not in (-1, beff.ndim - 1)
    if moved:
        beff = beff.movedim(dim, -1)
    U, Φ = _Beff2UPhi.apply(beff, γ2πdt)
    if moved:
        U = U.movedim(-1, dim)
    return U, Φ


@_host.half_via_float
def beff2ab(
    beff: Tensor, *,
    E1: Tensor = torch.tensor(0.), E2: Tensor = torch.tensor(0.),
    γ: Tensor = γH, dt: Tensor = dt0
) -> Tuple[Tensor, Tensor]:
    r"""Hargreaves' 𝐴/𝐵, mat/vec, of a pulse from its B-effectives (``beffective.py:40-104``;
    `doi:10.1002/mrm.1170 <https://doi.org/10.1002/mrm.1170>`_): after the pulse,
    ``M = A @ M0 + B`` (:func:`mrphy_amd.slowsims.blochsim_ab`).

    Usage:
        ``A, B = beff2ab(beff, *, E1, E2, γ, dt)``
    Inputs:
        - ``beff``: `(N, *Nd, nT, xyz)`, "Gauss", B-effective.
    Optionals (as the reference: the relaxation FACTORS, not T1/T2; default 0):
        - ``E1``, ``E2``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, ``exp(-dt/T1)``, ``exp(-dt/T2)``.
        - ``γ``: `()` ⊻ `(N ⊻ 1, *Nd ⊻ 1,)`, "Hz/Gauss"; ``dt``: `()` ⊻ `(N ⊻ 1,)`, "Sec".
    Outputs:
        - ``A``: `(N, *Nd, xyz, 3)`; ``B``: `(N, *Nd, xyz)`.

    One kernel carries the four columns of ``[I | 0]`` through the pulse (``Beff`` is read once).
    When ``beff`` requires grad the same kernel also records the 3x4 state before every step, and
    the adjoint of all four columns is one backward sweep over ``beff`` (``mrphy_beff2ab_bwd``), so
    gradients flow to ``beff`` -- and on to ``rf``, ``gr`` -- without a second look at ``beff``.  ``E1``,
    ``E2``, ``γ``, ``dt`` that require grad get theirs from the same sweep (round 4), as under the
    reference's autograd.

    Constants are rounded to ``beff``'s dtype first; outputs have ``beff``'s dtype (the reference
    silently promotes its result to fp64 when ``γ``/``dt`` are left at their fp64 defaults with
    fp32 ``beff``).
    """
    _host.require_device_tensor(beff, 'beff')
    device, dtype = beff.device, beff.dtype
    NNd = tuple(beff.shape[:-2])
    ndim = len(NNd)
    grad_on = torch.is_grad_enabled()
    consts_grad = grad_on and any(isinstance(x, Tensor) and x.requires_grad for x in (E1, E2, γ, dt))
    if consts_grad:
        # Round 4: the reference differentiates its plain torch ops w.r.t. E1, E2, γ, dt as well
        # (beffective.py:73-100).  Form γ2πdt and E1-1 with differentiable ops (its own expressions), let the
        # adjoint sweep return dL/d(γ2πdt), dL/dE1, dL/dE2, dL/d(E1-1) per spin (mrphy_beff2ab_bwd_consts), and
        # autograd chains them to the caller's tensors.
        E1, E2, γ, dt = (_host.pad_trailing(x.to(device=device, dtype=dtype), ndim) for x in (E1, E2, γ, dt))
    else:
        cdev = _host.const_device(device)
        E1, E2, γ, dt = (_host.pad_trailing(x.detach().to(device=cdev, dtype=dtype), ndim)
                         for x in (E1, E2, γ, dt))
    γ2πdt, E1_1 = 2 * π * γ * dt, E1 - 1            # beffective.py:73-74

    return _Beff2AB.apply(beff, γ2πdt, E1, E2, E1_1,
                          grad_on and (beff.requires_grad or consts_grad))


class _Beff2AB(Function):
    r"""``A, B = _Beff2AB.apply(beff, γ2πdt, E1, E2, E1_1, need_hist)``: ``mrphy_beff2ab`` and, when a
    gradient w.r.t. ``beff`` is wanted, ``mrphy_beff2ab_save`` + ``mrphy_beff2ab_bwd``: the adjoint of
    all four columns in one backward sweep over ``beff`` (the reference: autograd through its time
    loop, ``beffective.py:88-100``).  The forward numbers are the same with and without history."""

    @staticmethod
    def forward(ctx, beff, γ2πdt, E1, E2, E1_1, need_hist):
        lib = _lib.require_library()
        device, dtype = beff.device, beff.dtype
        NNd, nT = tuple(beff.shape[:-2]), beff.shape[-2]
        N, Nd = NNd[0], NNd[1:]
        nM = prod(Nd)
        g = _host.Bcast(γ2πdt, N, Nd, dtype, device)
        e1 = _host.Bcast(E1, N, Nd, dtype, device)
        e2 = _host.Bcast(E2, N, Nd, dtype, device)
        e1m1 = _host.Bcast(E1_1, N, Nd, dtype, device)
        assert (e1m1.sn, e1m1.sm) == (e1.sn, e1.sm)
        b = beff.detach().contiguous()
        A = torch.empty(NNd + (3, 3), dtype=dtype, device=device)
        B = torch.empty(NNd + (3,), dtype=dtype, device=device)
        code = _host.dtype_code(dtype, dtype)
        st = _host.current_stream(device)
        with torch.cuda.device(device):
            if need_hist:
                nb = int(lib.mrphy_beff2ab_hist_bytes(code, N, nM, nT))
                hist = torch.empty(max(nb, 16) // b.element_size(), dtype=dtype, device=device)
                rc = lib.mrphy_beff2ab_save(code, b.data_ptr(), *g.args, *e1.args, *e2.args,
                                            e1m1.t.data_ptr(), A.data_ptr(), B.data_ptr(),
                                            hist.data_ptr(), N, nM, nT, st)
            else:
                rc = lib.mrphy_beff2ab(code, b.data_ptr(), *g.args, *e1.args, *e2.args,
                                       e1m1.t.data_ptr(), A.data_ptr(), B.data_ptr(), N, nM, nT, st)
        _lib.check(rc, 'mrphy_beff2ab')
        if need_hist:
            ctx.save_for_backward(b, hist, g.t, e1.t, e2.t)
            ctx.meta = (code, (g.sn, g.sm), (e1.sn, e1.sm), (e2.sn, e2.sm), N, nM, nT, beff.dtype,
                        tuple((tuple(c.shape), c.dtype) for c in (γ2πdt, E1, E2, E1_1)), tuple(Nd))
        return A, B

    @staticmethod
    def backward(ctx, gA, gB):
        need = ctx.needs_input_grad
        if not any(need[:5]):
            return (None,) * 6
        from .sims import _reduce_to_const
        lib = _lib.require_library()
        b, hist, gt, e1t, e2t = ctx.saved_tensors
        code, gs, e1s, e2s, N, nM, nT, beff_dtype, cshapes, Nd = ctx.meta
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        gA = None if gA is None else gA.to(b.dtype).contiguous()
        gB = None if gB is None else gB.to(b.dtype).contiguous()
        gBeff = torch.empty_like(b)
        gcs = (None,) * 4
        with torch.cuda.device(b.device):
            if any(need[1:5]):
                gC = torch.empty((N * nM, 4), dtype=b.dtype, device=b.device)
                rc = lib.mrphy_beff2ab_bwd_consts(code, hist.data_ptr(), b.data_ptr(), gt.data_ptr(), *gs,
                                                  e1t.data_ptr(), *e1s, e2t.data_ptr(), *e2s, ptr(gA), ptr(gB),
                                                  gBeff.data_ptr(), gC.data_ptr(), N, nM, nT,
                                                  _host.current_stream(b.device))
                _lib.check(rc, 'mrphy_beff2ab_bwd_consts')
                full = gC.reshape((N,) + tuple(Nd) + (4,))
                gcs = tuple(_reduce_to_const(full[..., i], torch.empty(shp, dtype=dt_, device='meta'), N, Nd)
                            if want else None for i, ((shp, dt_), want) in enumerate(zip(cshapes, need[1:5])))
            else:
                rc = lib.mrphy_beff2ab_bwd(code, hist.data_ptr(), b.data_ptr(), gt.data_ptr(), *gs,
                                           e1t.data_ptr(), *e1s, e2t.data_ptr(), *e2s, ptr(gA), ptr(gB),
                                           gBeff.data_ptr(), N, nM, nT, _host.current_stream(b.device))
                _lib.check(rc, 'mrphy_beff2ab_bwd')
        return (gBeff.to(beff_dtype) if need[0] else None,) + gcs + (None,)


# the reference's __all__ spells it with U+03C6 (beffective.py:15); keep both names
globals()['beff2uφ'] = beff2uϕ
