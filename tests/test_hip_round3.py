r"""Round-3 additions to the GPU suite (``-m gpu``, through the C ABI):

* ``mobjs.Pulse.interpT`` as ``install()`` binds it (``mobjs.py:177-220``): the recorded config-5
  ``Pulse`` replayed through the bound method, bit for bit against the reference's output;
* BASELINE configs[4] in full (64^3 x 2048, every spin): gradients of both routes against exact
  differentiation (``oracle/bloch_c.c``), at the north star's 1e-5;
* a design iteration captured into a HIP graph and replayed: bit-identical gradients;
* ``slowsims.blochsim`` / ``blochsim_1step`` differentiable w.r.t. ``T1, T2, γ, dt`` (the four
  constants), as under the reference's autograd (``slowsims.py:86-112``); the paths that still are not
  (``slowsims.freeprec``, ``beff2ab``) raise instead of silently returning none.
"""
import os
import sys

import numpy as np
import pytest
import torch

import mrphy_amd
from mrphy_amd import beffective, sims, slowsims, synth
from util import golden, t, max_abs, rel_l2, record, to_dev

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(x):
    return None if x is None else x.to(DEV)


class PulseStandIn:
    r"""The attributes and constructor of ``mrphy.mobjs.Pulse`` that ``interpT`` and ``applypulse`` touch
    (``mobjs.py:56-125``: ``rf=None, gr=None, *, dt, gmax, smax, rfmax, desc, device, dtype``; a missing waveform is
    zeros; every tensor attribute is cast to the object's device / dtype; ``gmax/smax`` expand to `(N ⊻ 1, xyz)`,
    ``rfmax`` and ``dt`` 0-dim -> `(1,)`; ``shape``, ``is_cuda``).  The reference package does not exist on the
    GPU box; ``tests/test_abi_and_host.py::test_gpu_suite_stand_ins_mirror_the_reference_classes`` ties this class
    to the real one in the build container."""

    def __init__(self, rf=None, gr=None, *, dt=mrphy_amd.dt0, gmax=mrphy_amd.gmax0, smax=mrphy_amd.smax0,
                 rfmax=mrphy_amd.rfmax0, desc='generic pulse', device=torch.device('cpu'),
                 dtype=torch.float32):
        assert isinstance(device, torch.device) and isinstance(dtype, torch.dtype)
        assert not (rf is None and gr is None), "Missing both `rf` and `gr` inputs"
        kw = dict(device=device, dtype=dtype)
        self.device, self.dtype, self.is_cuda = device, dtype, device.type == 'cuda'
        if rf is None:
            rf = torch.zeros((gr.shape[0], 2, gr.shape[2]), **kw)
        elif gr is None:
            gr = torch.zeros((rf.shape[0], 3, rf.shape[2]), **kw)
        assert rf.shape[0] == gr.shape[0] and rf.shape[2] == gr.shape[2]
        self.shape = torch.Size((rf.shape[0], 1, rf.shape[2]))
        cast = lambda v: v.to(**kw) if isinstance(v, torch.Tensor) else torch.tensor(v, **kw)  # noqa: E731
        self.rf, self.gr = cast(rf), cast(gr)
        dt, gmax, smax, rfmax = cast(dt), cast(gmax), cast(smax), cast(rfmax)
        self.dt = dt[None] if dt.ndim == 0 else dt
        assert self.dt.ndim == 1
        self.gmax = gmax.expand((1 if gmax.ndim == 0 else gmax.shape[0], self.gr.shape[1]))
        self.smax = smax.expand((1 if smax.ndim == 0 else smax.shape[0], self.gr.shape[1]))
        self.rfmax = rfmax[None] if rfmax.ndim == 0 else (rfmax[:, 0] if rfmax.ndim == 2 and rfmax.shape[1] == 1 else rfmax)
        self.desc = desc

    def to(self, *, device=torch.device('cpu'), dtype=torch.float32):
        r"""``mobjs.Pulse.to`` (``mobjs.py:222-240``): the same waveforms, ``dt`` and ``desc`` on another device /
        dtype -- like the reference, WITHOUT the hardware limits (they fall back to the package defaults)."""
        if self.device == device and self.dtype == dtype:
            return self
        return PulseStandIn(self.rf, self.gr, dt=self.dt, desc=self.desc, device=device, dtype=dtype)


def test_pulse_interpT_bound_method_replays_config5():
    r"""The config-5 coarse ``Pulse`` (attributes recorded by make_golden.py next to the reference's
    own ``interpT`` output) through ``mrphy_amd._pulse_interpT`` -- the function ``install()`` binds
    to ``mobjs.Pulse.interpT`` -- on the device: waveforms and ``dt`` bit for bit, the reference's
    ``desc``, limits NOT carried over (``mobjs.py:219-220``), detached leaves (``mobjs.py:203``)."""
    I = golden('interp_f32')
    # (a device LEAF that requires grad, as a pulse under design is: .to() of a matching tensor is a no-op)
    coarse = PulseStandIn(dev(t(I['coarse_rf'])).requires_grad_(True), t(I['coarse_gr']), dt=t(I['coarse_dt']),
                          rfmax=torch.tensor(0.1), gmax=torch.tensor(2.0), desc=str(I['coarse_desc']),
                          device=DEV, dtype=torch.float32)
    fine = mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6], dtype=torch.float32))
    assert isinstance(fine, PulseStandIn) and fine.device == DEV and fine.dtype == torch.float32
    assert np.array_equal(fine.rf.cpu().numpy(), I['rf'])
    assert np.array_equal(fine.gr.cpu().numpy(), I['gr'])
    assert np.array_equal(fine.dt.cpu().numpy(), I['dt'])
    assert fine.desc == str(I['desc'])
    assert np.array_equal(fine.rfmax.cpu().numpy().reshape(-1), I['fine_rfmax'].reshape(-1)[:1])
    assert float(fine.gmax.reshape(-1)[0]) == float(I['fine_gmax'].reshape(-1)[0])
    assert not fine.rf.requires_grad and fine.rf.is_leaf          # graph cut, as in the reference
    # unchanged dwell time: a deep copy (mobjs.py:196-197)
    same = mrphy_amd._pulse_interpT(coarse, t(I['coarse_dt']))
    assert same is not coarse and torch.equal(same.rf, coarse.rf) and float(same.rfmax) == float(coarse.rfmax)
    # the one-tap kinds go through the device kernels too; dt with several entries asserts
    near = mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6], dtype=torch.float32), kind='nearest')
    assert near.rf.shape == (1, 2, 2048)
    with pytest.raises(AssertionError):
        mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6, 2e-6]))
    # scipy's spline kinds: the reference's own outputs for the same call (golden); the operator is
    # applied in fp64 on the device and rounded once, like scipy's fp64 result at mobjs.py:217
    for kind in ('slinear', 'quadratic', 'cubic'):
        f = mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6], dtype=torch.float32), kind=kind)
        assert max_abs(f.rf, I[f'{kind}_rf']) <= 1e-12 and max_abs(f.gr, I[f'{kind}_gr']) <= 1e-12, kind
    from mrphy_amd import interp
    y = torch.rand(1, 2, 64, dtype=torch.float64, device=DEV, requires_grad=True)
    gr64 = torch.rand(1, 3, 64, dtype=torch.float64, device=DEV)
    out = interp.interpT(y, gr64, torch.tensor([8e-6], dtype=torch.float64), torch.tensor([3e-6], dtype=torch.float64),
                         kind='cubic')[0]
    w = torch.rand_like(out)
    (out * w).sum().backward()
    W, _ = interp.interp_matrix(64, 8e-6, 3e-6, 'cubic')
    assert max_abs(y.grad, w @ torch.from_numpy(W).to(DEV)) < 1e-12           # the adjoint is W^T
    # opt-in differentiable form: install(interpT_graph=True)
    mrphy_amd._INTERP_GRAPH = True
    try:
        g = mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6], dtype=torch.float32))
        assert g.rf.requires_grad and max_abs(g.rf, I['rf']) == 0.0
        g.rf.sum().backward()
        assert coarse.rf.grad is not None and float(coarse.rf.grad.abs().sum()) > 0
    finally:
        mrphy_amd._INTERP_GRAPH = False


def test_slowsims_blochsim_differentiates_T1_T2_gamma_dt():
    r"""The reference's ``slowsims.blochsim`` forms ``E1, E2, γ2πdt`` with differentiable torch ops
    (``slowsims.py:86-98``): gradients w.r.t. ``T1, T2, γ, dt`` flow.  Here the adjoint sweep returns
    them (``mrphy_blochsim_bwd_consts``) -- per-spin ``T1``/``T2`` maps, a shared ``γ``, a one-entry
    ``dt`` -- against the oracle's autograd, with and without relaxation, fp64 (1e-9 relative) and fp32
    (1e-5); ``Mi`` / ``Beff`` gradients of the same call are unchanged by asking for the constants'."""
    import bloch_oracle as O
    for tag, dtype, tol in (('f64', torch.float64, 1e-9), ('f32', torch.float32, 2e-5)):
        n, nT = 5, 70                                   # 125 spins: tiles straddle; nT % 16 != 0
        sp = synth.cube_spins(n, dtype=dtype, seed_M0=3)
        p = synth.pulse(nT, dtype=dtype)
        beff = O.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        w = torch.cos(torch.arange(sp['M0'].numel(), dtype=torch.float64) * 0.37).reshape(sp['M0'].shape).to(dtype)
        for relax in (True, False):
            leaf = lambda x, d=None: (x.clone() if d is None else x.to(d).clone()).requires_grad_(True)  # noqa: E731
            names = ('T1', 'T2', 'γ', 'dt') if relax else ('γ', 'dt')
            ref = {k: leaf(sp[k] if k in sp else p[k]) for k in names}
            got = {k: leaf(sp[k] if k in sp else p[k], DEV) for k in names}
            Mo_r, Bo_r = leaf(sp['M0']), leaf(beff)
            Mo_h, Bo_h = leaf(sp['M0'], DEV), leaf(beff, DEV)
            (O.blochsim_slow(Mo_r, Bo_r, **ref) * w).sum().backward()
            with mrphy_amd.constants_on('native'):      # the oracle differentiates through torch.exp
                out = slowsims.blochsim(Mo_h, Bo_h, **got)
            (out * dev(w)).sum().backward()
            for k in names:
                a, b = got[k].grad, ref[k].grad
                assert a is not None and a.shape == b.shape, (tag, relax, k)
                e = record(f'const_grads.{tag}.{"relax" if relax else "norelax"}.{k}', rel_l2(a, b), tol)
                assert e <= tol, (tag, relax, k, e)
            assert rel_l2(Mo_h.grad, Mo_r.grad) <= tol and rel_l2(Bo_h.grad, Bo_r.grad) <= tol
            # the same Mi / Beff gradients as the call that does not ask for the constants'
            M2, B2 = leaf(sp['M0'], DEV), leaf(beff, DEV)
            (slowsims.blochsim(M2, B2, **{k: v.detach() for k, v in got.items()}) * dev(w)).sum().backward()
            assert rel_l2(M2.grad, Mo_h.grad) <= 1e-6 and rel_l2(B2.grad, Bo_h.grad) <= 1e-6


def test_constants_that_require_grad_elsewhere():
    r"""``sims.blochsim`` keeps the reference's contract (``None`` for ``T1, T2, γ, dt``,
    ``sims.py:154,269``)."""
    sp = synth.cube_spins(4, dtype=torch.float32, device=DEV)
    p = synth.pulse(32, dtype=torch.float32, device=DEV)
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    T1 = sp['T1'].clone().requires_grad_(True)
    kw = dict(T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    M0 = sp['M0'].clone().requires_grad_(True)
    b = sims.blochsim(M0, beff, T1=T1, **kw)           # the reference's own contract: T1.grad stays None
    b.sum().backward()
    assert T1.grad is None and M0.grad is not None
    with torch.no_grad():                              # nothing to differentiate: the plain kernels
        a = slowsims.blochsim(sp['M0'], beff, T1=T1, **kw)
    assert torch.equal(a, b.detach())
    # (round 4: slowsims.freeprec and beff2ab supply their constants' gradients too: test_hip_round4.py)


def test_config5_all_spins_gradients_vs_c_restatement():
    r"""BASELINE configs[4] at its real size -- all 262 144 spins x 2048 steps, the reference's own
    interpT output as the fine pulse -- forward + backward through both routes (rfgr2beff + blochsim
    with history + adjoints; fused K2 + K2b), in the product's DEFAULT constants mode, against
    ``oracle/bloch_c.c``: fp64 integration and differentiation of the same function on the same fp32
    field with the very constants the run used.  Bound: 1e-5 relative L2 on ``Mo, grad_M0, grad_rf,
    grad_gr`` (the reference tests gradient equality at atol 1e-4 in fp32, tests/test_sims.py:15,104-105)."""
    import bloch_c as C
    from mrphy_amd import fused
    I = golden('interp_f32')
    n, nT = 64, 2048
    pulse = dict(rf=t(I['rf']), gr=t(I['gr']), dt=t(I['dt']))
    spd = synth.cube_spins(n, dtype=torch.float32, device=DEV, seed_M0=2004)
    sp = {k: v.cpu() for k, v in spd.items()}
    with mrphy_amd.constants_on(None):            # default mode: exp in fp64, rounded once
        g, E1, E2, E1_1 = sims.relax_constants(spd['T1'], spd['T2'], spd['γ'], dev(pulse['dt']), 4, DEV)
    consts = dict(γ2πdt=g, E1=E1, E1_1=E1_1, E2=E2)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    cc = C.constants_from(g, E1, E2, E1_1, N=1, nM=n ** 3)
    Mo_e, gM0_e, grf_e, ggr_e = C.blochsim_rfgr_grad(sp['M0'], pulse['rf'], pulse['gr'], sp['loc'],
                                                     Δf=sp['Δf'], γ_beff=sp['γ'], consts=cc, field_f32=True)
    ex = dict(Mo=Mo_e, gM0=gM0_e, grf=grf_e, ggr=ggr_e)
    assert mrphy_amd.precision.get() == 'precise'
    for route in ('two', 'fused'):
        rf, gr = dev(pulse['rf']).requires_grad_(True), dev(pulse['gr']).requires_grad_(True)
        M0 = spd['M0'].clone().requires_grad_(True)
        if route == 'two':
            Mo = sims.blochsim_consts(M0, beffective.rfgr2beff(rf, gr, spd['loc'], Δf=spd['Δf'], γ=spd['γ']),
                                      **consts)
        else:
            Mo = fused.blochsim_rfgr(M0, rf, gr, spd['loc'], Δf=spd['Δf'], γ_beff=spd['γ'], consts=consts)
        Mo.sum().backward()
        got = dict(Mo=Mo, gM0=M0.grad, grf=rf.grad, ggr=gr.grad)
        for k in ('Mo', 'gM0', 'grf', 'ggr'):
            e = record(f'cfg5_all_spins.{route}.{k}.vs_exact', rel_l2(got[k], ex[k]), 1e-5)
            assert e <= 1e-5, (route, k, e)
        del Mo, got
        torch.cuda.empty_cache()


def test_hipgraph_capture_of_a_design_iteration():
    r"""A whole multi-scale design iteration -- interpT, fused forward with checkpoints, loss, fused
    adjoint, interpT adjoint -- captured into a HIP graph (``torch.cuda.CUDAGraph``) and replayed:
    the launches go to torch's current stream through the C ABI, allocate through torch and never
    synchronise, so stream capture sees all of them.  Replayed gradients are bit-identical to eager
    ones, also after the static inputs are updated in place (what an optimiser does)."""
    from mrphy_amd import fused, interp
    n, nT = 16, 256
    sp = synth.cube_spins(n, device=DEV)
    p = synth.pulse(nT // 2, device=DEV, dt=8e-6)
    dt_fine = torch.tensor([4e-6], device=DEV)
    rf = (0.05 * p['rf']).clone().requires_grad_(True)
    gr = p['gr'].clone().requires_grad_(True)

    def iteration():
        rf_f, gr_f, dt_f = interp.interpT(rf, gr, p['dt'], dt_fine)
        Mo = fused.blochsim_rfgr(sp['M0'], rf_f, gr_f, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'],
                                 T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=dt_f)
        return torch.autograd.grad((Mo ** 2).sum(), (rf, gr))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            a0, b0 = iteration()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a1, b1 = iteration()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    with torch.no_grad():
        rf.mul_(1.25)
        gr.add_(0.01)
    g.replay()
    torch.cuda.synchronize()
    a2, b2 = iteration()
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a0, a2)


@pytest.mark.parametrize('tag', ['f64', 'f32'])
@pytest.mark.parametrize('N,nM,nT,nC', [(1, 1, 1, 2), (2, 70, 37, 4), (1, 257, 257, 5), (1, 700, 300, 12),
                                        (2, 130, 64, 13), (1, 1030, 513, 24), (1, 66, 1000, 25), (1, 300, 96, 32)])
def test_multicoil_rfgr2beff_adjoint_shapes(tag, N, nM, nT, nC):
    r"""The parallel-transmit adjoint of ``rfgr2beff`` (2..32 coils; autograd over ``beffective.py:153-165``
    in the reference) on its own, against the same sums in fp64: every padded coil count of the
    step-per-thread pass (4 | 8 | 12 | 16 | 24 | 32, partly and completely filled), pulse lengths on
    either side of a 256-thread time tile (tail threads re-read the row's last time point and must store
    only their own), spin counts that leave a partial group of rows and a ragged last spin group, batch
    entries, and a ``grad_Beff`` that starts one element off a 16-byte boundary (a view into a larger
    buffer).  Run twice: the reduction is deterministic (bitwise)."""
    dt_ = torch.float64 if tag == 'f64' else torch.float32
    gen = torch.Generator().manual_seed(1000 * nC + nT)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64) * 2 - 1  # noqa: E731
    rf, gr = rnd(N, 2, nT, nC).to(dt_), rnd(N, 3, nT).to(dt_)
    loc, b1 = (rnd(N, nM, 3) * 6).to(dt_), rnd(N, nM, 2, nC).to(dt_)
    gB = rnd(N, nM, nT, 3).to(dt_)
    # fp64 sums of the very numbers the kernel reads
    G, B, Lc = gB.double(), b1.double(), loc.double()
    want_rf = torch.stack([torch.einsum('nstk,nskc->ntc', G[..., :2], B),
                           torch.einsum('nst,nsc->ntc', G[..., 1], B[:, :, 0]) -
                           torch.einsum('nst,nsc->ntc', G[..., 0], B[:, :, 1])], dim=1)
    want_gr = torch.einsum('nsi,nst->nit', Lc, G[..., 2])
    outs = []
    for rep in range(2):
        r, g = dev(rf).requires_grad_(True), dev(gr).requires_grad_(True)
        beff = beffective.rfgr2beff(r, g, dev(loc), b1Map=dev(b1))
        buf = torch.zeros(gB.numel() + 1, dtype=dt_, device=DEV)
        gview = buf[1:].view(gB.shape)                       # element-aligned only
        gview.copy_(gB)
        grf, ggr = torch.autograd.grad(beff, (r, g), gview)
        outs.append((grf, ggr))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    grf, ggr = outs[0]
    assert grf.shape == rf.shape and ggr.shape == gr.shape
    bound = 1e-12 if tag == 'f64' else 2e-6
    e_rf, e_gr = rel_l2(grf.cpu(), want_rf), rel_l2(ggr.cpu(), want_gr)
    record(f'k0adj.{tag}.N{N}_nM{nM}_nT{nT}_nC{nC}.grad_rf', e_rf, bound, 'multi-coil rfgr2beff adjoint vs fp64 sums')
    assert e_rf < bound and e_gr < bound, (e_rf, e_gr)


class SpinArrayStandIn:
    r"""The attributes and methods of ``mrphy.mobjs.SpinArray`` that ``applypulse`` touches (``mobjs.py:394-450``):
    compact ``M_, T1_, T2_, γ_`` `(N, nM[, xyz])`, the mask, ``extract`` / ``embed`` as ``install()`` binds them."""

    def __init__(self, mask, M_, T1_, T2_, γ_):
        self.device, self.dtype, self.mask = M_.device, M_.dtype, mask
        self.M_, self.T1_, self.T2_, self.γ_ = M_, T1_, T2_, γ_

    extract = mrphy_amd._spinarray_extract
    embed = mrphy_amd._spinarray_embed


@pytest.mark.parametrize('nC', [1, 2])
def test_applypulse_bound_method_is_the_fused_kernel(nC):
    r"""``mobjs.SpinArray.applypulse`` as ``install()`` binds it (``mrphy_amd._spinarray_applypulse``), on
    stand-ins for a device-resident ``SpinArray`` and a CPU ``Pulse`` (the reference's default: pulses are built
    on the CPU and moved by ``pulse2beff``, ``mobjs.py:651``): spatial ``loc / Δf / b1Map`` gathered through the
    mask as at ``mobjs.py:425-433``, ``doRelax`` off and on, ``doEmbed``, ``doUpdate`` -- each result BIT-IDENTICAL
    to ``rfgr2beff`` + ``sims.blochsim`` on the same arguments (what the reference's method composes,
    ``mobjs.py:435-446``), and the gradient to the pulse equal to the composed route's."""
    gen = torch.Generator().manual_seed(77 + nC)
    rnd = lambda *s: torch.rand(s, generator=gen)  # noqa: E731
    Nd, nT = (6, 5, 7), 48
    mask = (rnd(1, *Nd) > 0.3).to(DEV)
    nM = int(mask.sum())
    M_ = dev(rnd(1, nM, 3) * 2 - 1)
    T1_, T2_ = dev(0.5 + rnd(1, nM)), dev(0.02 + 0.1 * rnd(1, nM))
    γ_ = dev(torch.full((1, nM), 4257.6))
    arr = SpinArrayStandIn(mask, M_, T1_, T2_, γ_)
    rf = ((rnd(1, 2, nT, nC) if nC > 1 else rnd(1, 2, nT)) * 2 - 1) * 0.8
    pulse = PulseStandIn(rf, rnd(1, 3, nT) * 2 - 1, dt=torch.tensor(4e-6))            # on the CPU
    loc = dev((rnd(1, *Nd, 3) * 2 - 1) * 8)                  # spatial layout: gathered through the mask
    df = dev((rnd(1, *Nd) * 2 - 1) * 200)
    b1 = dev(rnd(1, *Nd, 2, nC) * 2 - 1) if nC > 1 else dev(rnd(1, *Nd, 2) * 2 - 1)
    loc_, df_, b1_ = (arr.extract(x) for x in (loc, df, b1))

    def composed(relax, rfd=None, grd=None):
        rfd = dev(pulse.rf) if rfd is None else rfd
        grd = dev(pulse.gr) if grd is None else grd
        beff = beffective.rfgr2beff(rfd, grd, loc_, Δf=df_, b1Map=b1_, γ=γ_)
        kw = dict(T1=T1_, T2=T2_) if relax else dict(T1=None, T2=None)
        return sims.blochsim(M_, beff, γ=γ_, dt=pulse.dt, **kw)
    with torch.no_grad():
        for relax in (True, False):
            got = mrphy_amd._spinarray_applypulse(arr, pulse, loc=loc, Δf=df, b1Map=b1, doRelax=relax)
            assert got.shape == (1, nM, 3) and torch.equal(got, composed(relax))
        got = mrphy_amd._spinarray_applypulse(arr, pulse, loc_=loc_, Δf_=df_, b1Map_=b1_, doEmbed=True, doUpdate=True)
        want = composed(True)
        assert arr.M_ is not M_ and torch.equal(arr.M_, want)            # doUpdate
        assert got.shape == (1, *Nd, 3) and torch.equal(arr.extract(got), want)   # doEmbed
        arr.M_ = M_
    with pytest.raises(AssertionError):
        mrphy_amd._spinarray_applypulse(arr, pulse, loc=loc, loc_=loc_)
    # gradient to the pulse: fused adjoint (nT % 16 == 0) vs the composed route
    pg = PulseStandIn(dev(pulse.rf).requires_grad_(True), dev(pulse.gr).requires_grad_(True), dt=dev(pulse.dt),
                      device=DEV)
    pg.rf, pg.gr = pg.rf.detach().requires_grad_(True), pg.gr.detach().requires_grad_(True)
    mrphy_amd._spinarray_applypulse(arr, pg, loc_=loc_, Δf_=df_, b1Map_=b1_).sum().backward()
    r2, g2 = dev(pulse.rf).requires_grad_(True), dev(pulse.gr).requires_grad_(True)
    composed(True, r2, g2).sum().backward()
    assert rel_l2(pg.rf.grad, r2.grad) < 2e-6 and rel_l2(pg.gr.grad, g2.grad) < 2e-6


def test_bench_three_ranks_rehearsed_on_one_gpu():
    r"""The N > 1 code path of ``bench.py`` on a box with one GPU: ``MRPHY_BENCH_REHEARSE=gloo python bench.py --gpus 3``
    -- the launcher starts three rank processes (before any GPU call), every rank simulates its block of the
    spin axis with the HIP kernels (a ragged split: 17^3 = 4913 spins over 3 ranks), the blocks are all-gathered
    (asynchronously, over gloo: RCCL refuses two ranks on one device), the clock is MAX-reduced, rank 0 prints the
    ONE JSON line.  Checked in the line: every rank's gathered copy is the same bit for bit, and each rank's own
    slice of it equals the fused kernel's result bit for bit.  (Three processes on the card: within the box's
    limit of six.)"""
    import json
    import subprocess
    env = dict(os.environ, MRPHY_BENCH_REHEARSE='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--cube', '17', '--nT', '64',
                        '--steps', '2', '--warmup', '1', '--no-cpu'], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, f'stdout must be the JSON line alone, got {len(lines)} lines: {r.stdout[:400]!r}'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 3 and len(d['per_rank_ms_per_step']) == 3 and d['rccl_ranks'] == 0
    assert 'rehearsal' in d and 'NOT a multi-GPU measurement' in d['rehearsal']
    assert d['gathered_result_identical_on_all_ranks'] is True
    assert d['kernels']['K2_fused_rfgr_fwd']['equals_K0_K1_bitwise'] is True
    assert d['config']['spins'] == 17 ** 3 and d['config']['parallelism'] == 'spins/3'
