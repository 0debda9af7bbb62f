r"""Round-3 additions to the GPU suite (``-m gpu``, through the C ABI):

* ``mobjs.Pulse.interpT`` as ``install()`` binds it (``mobjs.py:177-220``): the recorded config-5
  ``Pulse`` replayed through the bound method, bit for bit against the reference's output;
* boundary error behaviour: ``T1/T2/γ/dt`` that require grad raise instead of silently getting
  none (the reference's ``slowsims`` would have differentiated them, ``slowsims.py:86-98``).
"""
import os

import numpy as np
import pytest
import torch

import mrphy_amd
from mrphy_amd import beffective, sims, slowsims, synth
from util import golden, t, max_abs

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(x):
    return None if x is None else x.to(DEV)


class PulseStandIn:
    r"""The attributes and constructor signature of ``mrphy.mobjs.Pulse`` that ``interpT`` touches
    (``mobjs.py:56-99``: ``rf, gr, *, dt, gmax, smax, rfmax, desc, device, dtype``; limits default
    to the package constants).  The reference package does not exist on the GPU box."""

    def __init__(self, rf, gr, *, dt=mrphy_amd.dt0, gmax=mrphy_amd.gmax0, smax=mrphy_amd.smax0,
                 rfmax=mrphy_amd.rfmax0, desc='generic pulse', device=torch.device('cpu'),
                 dtype=torch.float32):
        kw = dict(device=device, dtype=dtype)
        self.device, self.dtype = device, dtype
        self.rf, self.gr = rf.to(**kw), gr.to(**kw)
        dt = dt.to(**kw)
        self.dt = dt[None] if dt.ndim == 0 else dt
        self.gmax, self.smax, self.rfmax = (torch.as_tensor(x).to(**kw) for x in (gmax, smax, rfmax))
        self.desc = desc


def test_pulse_interpT_bound_method_replays_config5():
    r"""The config-5 coarse ``Pulse`` (attributes recorded by make_golden.py next to the reference's
    own ``interpT`` output) through ``mrphy_amd._pulse_interpT`` -- the function ``install()`` binds
    to ``mobjs.Pulse.interpT`` -- on the device: waveforms and ``dt`` bit for bit, the reference's
    ``desc``, limits NOT carried over (``mobjs.py:219-220``), detached leaves (``mobjs.py:203``)."""
    I = golden('interp_f32')
    coarse = PulseStandIn(t(I['coarse_rf']).requires_grad_(True), t(I['coarse_gr']), dt=t(I['coarse_dt']),
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
    # opt-in differentiable form: install(interpT_graph=True)
    mrphy_amd._INTERP_GRAPH = True
    try:
        g = mrphy_amd._pulse_interpT(coarse, torch.tensor([4e-6], dtype=torch.float32))
        assert g.rf.requires_grad and max_abs(g.rf, I['rf']) == 0.0
        g.rf.sum().backward()
        assert coarse.rf.grad is not None and float(coarse.rf.grad.abs().sum()) > 0
    finally:
        mrphy_amd._INTERP_GRAPH = False


def test_constants_that_require_grad_raise():
    r"""The kernels differentiate w.r.t. ``Mi`` and ``Beff`` (``sims.py:27,149-150``).  The
    reference's ``slowsims`` forms ``E1, E2, γ2πdt`` with differentiable torch ops
    (``slowsims.py:86-98``, ``beffective.py:88-100``), so there ``T1/T2/γ/dt`` get gradients: a
    caller asking for them here is told so instead of silently receiving none."""
    sp = synth.cube_spins(4, dtype=torch.float32, device=DEV)
    p = synth.pulse(32, dtype=torch.float32, device=DEV)
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    T1 = sp['T1'].clone().requires_grad_(True)
    kw = dict(T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    for fn in (slowsims.blochsim, sims.blochsim):
        with pytest.raises(RuntimeError, match='T1'):
            fn(sp['M0'], beff, T1=T1, **kw)
        with torch.no_grad():                          # nothing to differentiate: fine
            fn(sp['M0'], beff, T1=T1, **kw)
    γ = sp['γ'].clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match='γ'):
        slowsims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=γ, dt=p['dt'])
    E1 = torch.exp(-p['dt'] / sp['T1']).requires_grad_(True)
    with pytest.raises(RuntimeError, match='E1'):
        beffective.beff2ab(beff, E1=E1, E2=torch.exp(-p['dt'] / sp['T2']), γ=sp['γ'], dt=p['dt'])
