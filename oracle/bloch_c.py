r"""ctypes front end of ``oracle/bloch_c.c`` (plain C, fp64, OpenMP).  TEST INFRASTRUCTURE: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import it.

Inputs are CPU tensors of any float dtype; everything is computed in double from their values.
Per-spin constants are formed here exactly as the reference does (``sims.py:62,74-76``) from
``T1, T2, γ, dt`` unless passed in.
"""
import ctypes
import os
import subprocess
from math import pi as π

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, 'libbloch_c.so')
_lib = None
_dp, _i64 = ctypes.c_void_p, ctypes.c_int64


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'bloch_c.c')
        if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
            subprocess.run(['make', '-s', '-C', _HERE], check=True)
        _lib = ctypes.CDLL(_PATH)
        _lib.oracle_rfgr2beff_f64.argtypes = [_dp, _i64, _dp, _i64, _dp, _dp, _dp, _dp] + [_i64] * 4
        _lib.oracle_blochsim_f64.argtypes = [_dp] * 7 + [_i64] * 2
        _lib.oracle_blochsim_rfgr_f64.argtypes = [_dp, _dp, _i64, _dp, _i64] + [_dp] * 8 + [_i64] * 4
        _lib.oracle_blochsim_rfgr_f32field.argtypes = _lib.oracle_blochsim_rfgr_f64.argtypes
        _lib.oracle_blochsim_bwd_f64.argtypes = [_dp] * 9 + [_i64] * 2
        _lib.oracle_blochsim_rfgr_grad.argtypes = ([_dp, _dp, _i64, _dp, _i64] + [_dp] * 12 + [_i64] * 4
                                                   + [ctypes.c_int])
        for f in (_lib.oracle_rfgr2beff_f64, _lib.oracle_blochsim_f64, _lib.oracle_blochsim_rfgr_f64,
                  _lib.oracle_blochsim_rfgr_f32field, _lib.oracle_blochsim_bwd_f64,
                  _lib.oracle_blochsim_rfgr_grad):
            f.restype = None
    return _lib


def _d(x):
    return x.detach().to('cpu', torch.float64).contiguous()


def _rows(x, N, nM):
    r"""`()` ⊻ `(N ⊻ 1, nM ⊻ 1[, 1...])` -> contiguous `(N*nM,)` double."""
    x = _d(x)
    x = x.reshape(x.shape[:2]) if x.ndim > 2 else x
    while x.ndim < 2:
        x = x[None]
    return x.expand(N, nM).contiguous().reshape(-1)


def _p(x):
    return None if x is None else x.data_ptr()


def _nm(x):
    r"""`()`, `(N ⊻ 1,)` (dt) or `(N ⊻ 1, nM ⊻ 1[, 1...])` -> a 2-D double tensor `(N ⊻ 1, nM ⊻ 1)`."""
    x = _d(x)
    if x.ndim == 0:
        return x.reshape(1, 1)
    if x.ndim == 1:
        return x.reshape(-1, 1)
    return x.reshape(x.shape[0], -1)


def constants(T1, T2, γ, dt, N, nM):
    r"""``(γ2πdt, E1, E2, E1-1)`` per spin, in double, by the reference's expressions
    (``sims.py:62,74-76``); ``T1 = T2 = None``: no relaxation."""
    g = _rows(2 * π * _nm(γ) * _nm(dt), N, nM)
    if T1 is None:
        return g, None, None, None
    E1, E2 = torch.exp(-_nm(dt) / _nm(T1)), torch.exp(-_nm(dt) / _nm(T2))
    return g, _rows(E1, N, nM), _rows(E2, N, nM), _rows(E1 - 1, N, nM)


def constants_from(γ2πdt, E1=None, E2=None, E1_1=None, *, N, nM):
    r"""The same tuple from constants somebody else computed (e.g. the fp32 ones a GPU run used)."""
    f = lambda x: None if x is None else _rows(x, N, nM)  # noqa: E731
    return f(γ2πdt), f(E1), f(E2), f(E1_1)


def _pulse(rf, gr, loc, Δf, b1Map, γ):
    loc = _d(loc)
    N, nM = loc.shape[0], loc.shape[1]
    rf, gr = _d(rf), _d(gr)
    if b1Map is None and rf.ndim == 4:
        rf = rf.sum(dim=-1)
    rf4 = rf if rf.ndim == 4 else rf[..., None].contiguous()
    nT, nC = gr.shape[2], rf4.shape[-1]
    b1 = None
    if b1Map is not None:
        b1 = _d(b1Map)
        b1 = b1 if b1.ndim == 4 else b1[..., None]
        b1 = b1.expand(N, nM, 2, nC).contiguous()
    dfg = None if Δf is None else (_rows(Δf, N, nM) / _rows(γ, N, nM)).contiguous()
    rf_sn = rf4.stride(0) if (rf4.shape[0] == N and N > 1) else 0
    gr_sn = gr.stride(0) if (gr.shape[0] == N and N > 1) else 0
    return loc, rf4.contiguous(), rf_sn, gr, gr_sn, dfg, b1, N, nM, nT, nC


def rfgr2beff(rf, gr, loc, *, Δf=None, b1Map=None, γ=torch.tensor(4257.6, dtype=torch.float64)):
    lib = _load()
    loc, rf4, rf_sn, gr, gr_sn, dfg, b1, N, nM, nT, nC = _pulse(rf, gr, loc, Δf, b1Map, γ)
    beff = torch.empty((N, nM, nT, 3), dtype=torch.float64)
    lib.oracle_rfgr2beff_f64(rf4.data_ptr(), rf_sn, gr.data_ptr(), gr_sn, loc.data_ptr(), _p(dfg),
                             _p(b1), beff.data_ptr(), N, nM, nT, nC)
    return beff


def field_f32(rf, gr, loc, *, Δf=None, b1Map=None, γ_beff=torch.tensor(4257.6, dtype=torch.float64)):
    r"""The single-precision field the ``field_f32=True`` integrations use, as an fp32 tensor `(N, nM, nT, 3)`: pinned
    bit for bit to the reference's own fp32 ``rfgr2beff`` rows (``tests/golden/big_beff_rows_f32.npz``)."""
    import ctypes
    lib = _load()
    assert all(x is None or x.dtype == torch.float32 for x in (rf, gr, loc, Δf, b1Map))
    loc, rf4, rf_sn, gr, gr_sn, dfg, b1, N, nM, nT, nC = _pulse(rf, gr, loc, Δf, b1Map, γ_beff)
    if Δf is not None:
        dfg = (_rows(Δf, N, nM).float() / _rows(γ_beff.float(), N, nM).float()).double().contiguous()
    beff = torch.empty((N, nM, nT, 3), dtype=torch.float32)
    fn = lib.oracle_field_f32
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 4
    fn(rf4.data_ptr(), rf_sn, gr.data_ptr(), gr_sn, loc.data_ptr(), _p(dfg), _p(b1), beff.data_ptr(), N, nM, nT, nC)
    return beff


def blochsim(Mi, Beff, *, T1=None, T2=None, γ=torch.tensor(4257.6, dtype=torch.float64),
             dt=torch.tensor(4e-6, dtype=torch.float64), consts=None):
    lib = _load()
    Mi, Beff = _d(Mi), _d(Beff)
    N, nM, nT = Beff.shape[0], Beff.shape[1], Beff.shape[2]
    g, E1, E2, E1m1 = consts if consts is not None else constants(T1, T2, γ, dt, N, nM)
    Mo = torch.empty_like(Mi)
    lib.oracle_blochsim_f64(Mi.data_ptr(), Beff.data_ptr(), g.data_ptr(), _p(E1), _p(E2), _p(E1m1),
                            Mo.data_ptr(), N * nM, nT)
    return Mo


def blochsim_rfgr(Mi, rf, gr, loc, *, Δf=None, b1Map=None, γ_beff=torch.tensor(4257.6, dtype=torch.float64),
                  T1=None, T2=None, γ=torch.tensor(4257.6, dtype=torch.float64),
                  dt=torch.tensor(4e-6, dtype=torch.float64), consts=None, field_f32=False):
    r"""``blochsim(Mi, rfgr2beff(...))`` in double.  ``field_f32=True`` (float inputs): every step's
    field is first formed in single precision exactly as the reference forms its fp32 ``Beff``
    tensor, then integrated in double -- exact arithmetic on the same fp32 field."""
    lib = _load()
    Mi = _d(Mi)
    if field_f32:
        assert all(x is None or x.dtype == torch.float32 for x in (rf, gr, loc, Δf, b1Map)), \
            'field_f32 reproduces the fp32 field: pass float32 inputs'
    loc, rf4, rf_sn, gr, gr_sn, dfg, b1, N, nM, nT, nC = _pulse(rf, gr, loc, Δf, b1Map, γ_beff)
    g, E1, E2, E1m1 = consts if consts is not None else constants(T1, T2, γ, dt, N, nM)
    Mo = torch.empty_like(Mi)
    fn = lib.oracle_blochsim_rfgr_f64
    if field_f32:
        if Δf is not None:     # df/gamma divided in single precision, as the kernels and ATen do
            dfg = (_rows(Δf, N, nM).float() / _rows(γ_beff.float(), N, nM).float()).double().contiguous()
        fn = lib.oracle_blochsim_rfgr_f32field
    fn(Mi.data_ptr(), rf4.data_ptr(), rf_sn, gr.data_ptr(), gr_sn, loc.data_ptr(), _p(dfg), _p(b1),
       g.data_ptr(), _p(E1), _p(E2), _p(E1m1), Mo.data_ptr(), N, nM, nT, nC)
    return Mo


def blochsim_bwd(Mi, Beff, grad_Mo, *, T1=None, T2=None, γ=torch.tensor(4257.6, dtype=torch.float64),
                 dt=torch.tensor(4e-6, dtype=torch.float64), consts=None):
    r"""``(grad_Mi, grad_Beff)`` of ``blochsim`` for the cotangent ``grad_Mo``: the explicit adjoint
    of ``sims.py:135-269`` in double (forward recomputed inside)."""
    lib = _load()
    Mi, Beff, gMo = _d(Mi), _d(Beff), _d(grad_Mo)
    N, nM, nT = Beff.shape[0], Beff.shape[1], Beff.shape[2]
    g, E1, E2, E1m1 = consts if consts is not None else constants(T1, T2, γ, dt, N, nM)
    gMi, gB = torch.empty_like(Mi), torch.empty_like(Beff)
    lib.oracle_blochsim_bwd_f64(Mi.data_ptr(), Beff.data_ptr(), g.data_ptr(), _p(E1), _p(E2), _p(E1m1),
                                gMo.data_ptr(), gMi.data_ptr(), gB.data_ptr(), N * nM, nT)
    return gMi, gB


def blochsim_rfgr_grad(Mi, rf, gr, loc, grad_Mo=None, *, Δf=None, b1Map=None,
                       γ_beff=torch.tensor(4257.6, dtype=torch.float64), T1=None, T2=None,
                       γ=torch.tensor(4257.6, dtype=torch.float64),
                       dt=torch.tensor(4e-6, dtype=torch.float64), consts=None, field_f32=False):
    r"""``Mo, grad_Mi, grad_rf, grad_gr`` of ``L = <grad_Mo, blochsim(Mi, rfgr2beff(rf, gr, ...))>``
    in double (``grad_Mo`` defaults to ones: ``L = Mo.sum()``), without materialising ``Beff``.
    ``grad_rf`` is ``(N, 2, nT[, nC])`` in the layout of the ``rf`` that ``rfgr2beff`` consumes (a
    multi-coil ``rf`` without a ``b1Map`` is summed over coils first, as there), ``grad_gr``
    ``(N, 3, nT)``: one entry per batch element (sum them for a broadcast pulse).
    ``field_f32``: as :func:`blochsim_rfgr`."""
    lib = _load()
    Mi = _d(Mi)
    if field_f32:
        assert all(x is None or x.dtype == torch.float32 for x in (rf, gr, loc, Δf, b1Map)), \
            'field_f32 reproduces the fp32 field: pass float32 inputs'
    four = rf.ndim == 4 and b1Map is not None
    loc, rf4, rf_sn, gr, gr_sn, dfg, b1, N, nM, nT, nC = _pulse(rf, gr, loc, Δf, b1Map, γ_beff)
    if field_f32 and Δf is not None:
        dfg = (_rows(Δf, N, nM).float() / _rows(γ_beff.float(), N, nM).float()).double().contiguous()
    g, E1, E2, E1m1 = consts if consts is not None else constants(T1, T2, γ, dt, N, nM)
    gMo = torch.ones_like(Mi) if grad_Mo is None else _d(grad_Mo)
    Mo, gMi = torch.empty_like(Mi), torch.empty_like(Mi)
    grf = torch.empty((N, 2, nT, nC), dtype=torch.float64)
    ggr = torch.empty((N, 3, nT), dtype=torch.float64)
    lib.oracle_blochsim_rfgr_grad(Mi.data_ptr(), rf4.data_ptr(), rf_sn, gr.data_ptr(), gr_sn,
                                  loc.data_ptr(), _p(dfg), _p(b1), g.data_ptr(), _p(E1), _p(E2),
                                  _p(E1m1), gMo.data_ptr(), Mo.data_ptr(), gMi.data_ptr(),
                                  grf.data_ptr(), ggr.data_ptr(), N, nM, nT, nC, int(bool(field_f32)))
    return Mo, gMi, (grf if four else grf[..., 0]), ggr
