r"""``freeprec``, on-device ``Pulse.interpT``, mask gather / scatter and ``_update_loc_``, ``beff2ab`` / ``blochsim_ab`` -- forward,
adjoints and the gradients the reference's autograd supplies.

Regrouped by component in round 5 from ``test_hip_parity.py`` / ``test_hip_round{2,3,4}.py`` (no assertion changed; each test keeps its name).
"""
import pytest

from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_freeprec(tag):
    r"""sims.freeprec / slowsims.freeprec forward and grad_Mi vs the reference's golden outputs
    (both of its implementations), its known answer, and the live oracle on a bigger case."""
    G = golden(f'freeprec_{tag}')
    known = np.array([[[0., -0.5, 0.5], [-0.5, 0, 0.5], [0., 0., 1.]]])
    for name, kw in cases.freeprec_variants(DT[tag]).items():
        kw = to_dev(dict(kw), DEV)
        M, dur = kw.pop('M'), kw.pop('dur')
        for fn in (sims.freeprec, slowsims.freeprec):
            Mi = M.clone().requires_grad_(True)
            Mo = fn(Mi, dur, **kw)
            w = torch.cos(torch.arange(Mo.numel(), dtype=torch.float64) * 0.53).reshape(Mo.shape)
            (Mo * w.to(device=DEV, dtype=DT[tag])).sum().backward()
            assert max_abs(Mi, M) == 0.0 and Mo.data_ptr() != Mi.data_ptr()
            for impl in ('sims', 'slow'):
                assert_close(Mo, G[f'{name}.Mo_{impl}'], tag, f'{name}.Mo vs {impl}')
                assert_close(Mi.grad, G[f'{name}.gMi_{impl}'], tag, f'{name}.gMi vs {impl}')
            if name == 'known':
                assert max_abs(Mo, known) < (1e-9 if tag == 'f64' else 1e-6)
    # ragged size with general *Nd, per-spin everything
    gen = torch.Generator().manual_seed(3)
    M = torch.rand((2, 9, 11, 3), generator=gen, dtype=torch.float64).to(DT[tag])
    T1 = (0.5 + torch.rand((2, 9, 11), generator=gen, dtype=torch.float64)).to(DT[tag])
    T2 = (0.02 + 0.1 * torch.rand((2, 9, 11), generator=gen, dtype=torch.float64)).to(DT[tag])
    df = ((torch.rand((2, 9, 11), generator=gen, dtype=torch.float64) * 2 - 1) * 500).to(DT[tag])
    dur = torch.tensor([2e-3, 5e-3], dtype=DT[tag])
    want = O.freeprec(M, dur, T1=T1, T2=T2, Δf=df)
    got = sims.freeprec(dev(M), dev(dur), T1=dev(T1), T2=dev(T2), Δf=dev(df))
    assert got.shape == M.shape
    assert_close(got, want, tag, 'general Nd')
    assert sims.freeprec(torch.zeros(1, 0, 3, device=DEV), dev(dur[:1])).shape == (1, 0, 3)


@pytest.mark.usefixtures('host_constants')
def test_interpT_on_device():
    r"""Pulse.interpT(kind='linear') on the device: the reference's known answer
    (tests/test_mobjs.py:160-195), its output for the config-5 coarse pulse (golden, bit for bit),
    the 255-sample quirk, a multi-coil rf, and the adjoint against a dense interpolation matrix."""
    from mrphy_amd.interp import interpT, interp_grid
    f64 = torch.float64
    nT = 11
    lin = lambda a, b: torch.linspace(a, b, nT, dtype=f64).reshape(1, 1, nT)  # noqa: E731
    rf = 0.1 * torch.cat([lin(0., 1.), lin(1., 0.)], 1)
    gr = 0.1 * torch.cat([lin(0., 1.), lin(1., 0.), torch.ones(1, 1, nT, dtype=f64)], 1)
    dt = torch.tensor([4e-6], dtype=f64)
    rf_n, gr_n, dt_n = interpT(dev(rf), dev(gr), dev(dt), dev(dt * 5))
    assert max_abs(rf_n, np.array([[[0.04, 0.09], [0.06, 0.01]]])) < 1e-9
    assert max_abs(gr_n, np.array([[[0.04, 0.09], [0.06, 0.01], [0.1, 0.1]]])) < 1e-9
    assert float(dt_n) == float(dt * 5)
    same = interpT(dev(rf), dev(gr), dev(dt), dev(dt.clone()))
    assert same[0].data_ptr() == dev(rf).data_ptr() or max_abs(same[0], rf) == 0.0
    # config 5: coarse 1024 @ 8e-6 -> 4e-6, fp32: exactly what the reference produced
    I = golden('interp_f32')
    p = synth.pulse(1024, dtype=torch.float32, dt=8e-6)
    rf5, gr5, dt5 = interpT(dev(p['rf']), dev(p['gr']), dev(p['dt']), torch.tensor([4e-6], dtype=torch.float32))
    assert rf5.shape == (1, 2, 2048) and rf5.dtype == torch.float32
    # (the coarse pulse is re-synthesised here: torch.sin near π differs by ~5e-20 between hosts)
    assert max_abs(rf5, I['rf']) < 1e-12 and max_abs(gr5, I['gr']) < 1e-12 and max_abs(dt5, I['dt']) == 0.0
    q = synth.pulse(512, dtype=torch.float32, dt=4e-6)
    assert interpT(dev(q['rf']), dev(q['gr']), dev(q['dt']), torch.tensor(8e-6, dtype=f64))[0].shape[2] == 255
    # multi-coil rf (time is axis 2 of 4) and the adjoint
    gen = torch.Generator().manual_seed(9)
    rfc = torch.rand((2, 2, 37, 3), generator=gen, dtype=f64).requires_grad_(True)
    grc = torch.rand((2, 3, 37), generator=gen, dtype=f64).requires_grad_(True)
    lo, w, dx, n = interp_grid(37, 4e-6, 1.5e-6)
    W = torch.zeros(n, 38, dtype=f64)                       # dense map on the zero-prepended source
    for j in range(n):
        W[j, lo[j] + 1] += w[j] / dx[j]
        W[j, lo[j]] += 1 - w[j] / dx[j]
    ext = lambda x: torch.cat([torch.zeros_like(x[..., :1]), x], dim=-1)  # noqa: E731
    want_rf = (ext(rfc.movedim(2, -1)) @ W.T).movedim(-1, 2)
    want_gr = ext(grc) @ W.T
    cw = torch.cos(torch.arange(want_rf.numel(), dtype=f64)).reshape(want_rf.shape)
    cg = torch.sin(torch.arange(want_gr.numel(), dtype=f64)).reshape(want_gr.shape)
    ((want_rf * cw).sum() + (want_gr * cg).sum()).backward()
    rfd, grd = dev(rfc.detach()).requires_grad_(True), dev(grc.detach()).requires_grad_(True)
    got_rf, got_gr, _ = interpT(rfd, grd, torch.tensor([4e-6], dtype=f64), torch.tensor([1.5e-6], dtype=f64))
    assert got_rf.shape == want_rf.shape and got_gr.shape == want_gr.shape
    assert max_abs(got_rf, want_rf) < 1e-12 and max_abs(got_gr, want_gr) < 1e-12
    ((got_rf * dev(cw)).sum() + (got_gr * dev(cg)).sum()).backward()
    assert max_abs(rfd.grad, rfc.grad) < 1e-12 and max_abs(grd.grad, grc.grad) < 1e-12


# ---------------------------------------------------------------------------------------------
# mask gather/scatter (SpinArray.extract/embed) and SpinCube._update_loc_
# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_masks_golden(tag):
    r"""Bit-exact against the reference's outputs: extract, embed (fresh: NaN outside the mask;
    `out=`: untouched outside), their gradients, and the cube locations."""
    from mrphy_amd import masks
    G, c = golden(f'masks_{tag}'), cases.mask_case(DT[tag])
    ix = masks.MaskIndex(dev(c['mask']))
    assert (ix.nM, ix.nV, ix.Nd) == (c['nM'], c['nV'], c['Nd'])
    for name, v in c['spatial'].items():
        assert np.array_equal(masks.extract(dev(v), ix).cpu().numpy(), G[f'extract.{name}'])
    for name, v_ in c['compact'].items():
        assert np.array_equal(masks.embed(dev(v_), ix).cpu().numpy(), G[f'embed.{name}'],
                              equal_nan=True)
    base = dev(c['spatial']['M'].clone())
    got = masks.embed(dev(c['compact']['M']), dev(c['mask']), out=base)      # mask given directly
    assert got.data_ptr() == base.data_ptr()
    assert np.array_equal(got.cpu().numpy(), G['embed_out.M'])
    out_ = torch.empty((c['N'], c['nM'], 3), dtype=DT[tag], device=DEV)
    assert masks.extract(dev(c['spatial']['M']), ix, out_=out_).data_ptr() == out_.data_ptr()
    assert np.array_equal(out_.cpu().numpy(), G['extract.M'])
    assert np.array_equal(masks.cube_loc(ix, dev(c['fov']), dev(c['ofst'])).cpu().numpy(), G['loc_'])
    # gradients (extract and embed are each other's adjoints)
    v = dev(c['spatial']['M']).requires_grad_(True)
    w_ = ((torch.arange(c['N'] * c['nM'] * 3, dtype=torch.float64) * 7) % 33 - 16) \
        .reshape(c['N'], c['nM'], 3).to(DT[tag])
    (masks.extract(v, ix) * dev(w_)).sum().backward()
    assert np.array_equal(v.grad.cpu().numpy(), G['extract.gM'])
    v_ = dev(c['compact']['M']).requires_grad_(True)
    w = ((torch.arange(v.numel(), dtype=torch.float64) * 5) % 29 - 14).reshape(v.shape).to(DT[tag])
    torch.nan_to_num(masks.embed(v_, ix) * dev(w)).sum().backward()
    assert np.array_equal(v_.grad.cpu().numpy(), G['embed.gM_'])
    # the reference's own mobjs test case (test_mobjs.py:98-131): its cube's loc_
    M = golden(f'mobjs_{tag}')
    fov = torch.tensor([[3., 3., 3.]], dtype=DT[tag], device=DEV)
    ofst = torch.tensor([[0., 0., 1.]], dtype=DT[tag], device=DEV)
    loc_ = masks.cube_loc(dev(torch.from_numpy(M['mask'])), fov, ofst)
    assert np.array_equal(loc_.cpu().numpy(), M['loc_'])


@pytest.mark.usefixtures('host_constants')
def test_masks_properties_and_edges():
    r"""Size-independent properties at a 96^3 grid (random mask), vs the oracle, and edge cases."""
    from mrphy_amd import masks
    g = torch.Generator().manual_seed(5)
    Nd, N = (96, 96, 96), 2
    mask = (torch.rand((1,) + Nd, generator=g) < 0.6)
    ix = masks.MaskIndex(dev(mask))
    assert ix.nM == int(mask.sum())
    v = torch.randn((N,) + Nd + (3,), generator=g)
    v_ = masks.extract(dev(v), ix)
    assert torch.equal(v_.cpu(), O.mask_extract(v, mask))                  # vs the oracle
    back = masks.embed(v_, ix)                                             # round trip
    inside = mask.expand((N,) + Nd)
    assert torch.equal(back.cpu()[inside], v[inside]) and bool(torch.isnan(back.cpu()[~inside]).all())
    assert torch.equal(masks.extract(back, ix), v_)                        # idempotent
    fov = torch.tensor([[24., 24., 12.], [20., 22., 7.]])
    ofst = torch.tensor([[0., 1., -2.], [0.5, 0., 0.]])
    assert torch.equal(masks.cube_loc(ix, dev(fov), dev(ofst)).cpu(), O.cube_loc(mask, fov, ofst))
    # synth.cube_spins' grid is the same construction: FOV*(i - n//2)/n
    # edge cases: full mask, empty mask, one voxel, odd sizes
    for m in (torch.ones((1, 3, 1, 5), dtype=torch.bool), torch.zeros((1, 2, 3, 4), dtype=torch.bool),
              torch.ones((1, 1, 1, 1), dtype=torch.bool)):
        ixm = masks.MaskIndex(dev(m))
        x = torch.randn((2,) + tuple(m.shape[1:]) + (2,), generator=g, dtype=torch.float64)
        xe = masks.extract(dev(x), ixm)
        assert xe.shape == (2, int(m.sum()), 2) and torch.equal(xe.cpu(), O.mask_extract(x, m))
        xb = masks.embed(xe, ixm)
        assert xb.shape == x.shape
        assert np.array_equal(xb.cpu().numpy(), O.mask_embed(xe.cpu(), m).numpy(), equal_nan=True)
        f = torch.ones((2, 3), dtype=torch.float64)
        assert torch.equal(masks.cube_loc(ixm, dev(f), dev(f)).cpu(), O.cube_loc(m, f, f))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        masks.extract(v, ix)
    with pytest.raises(AssertionError):
        masks.extract(dev(v[:, :5]), ix)


# ---------------------------------------------------------------------------------------------
# Hargreaves A/B -- beffective.beff2ab + slowsims.blochsim_ab
# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_ab_reference_case(tag):
    r"""The reference's own check (test_slowsims.py:64-96): A, B of the 3-spin case, Mo3 = A M0 + B
    against the known answer, and the gradient chain Mo3 -> A, B -> beff -> rf, gr."""
    G, c = golden(f'ab3_{tag}'), to_dev(cases.ref_case(3, DT[tag]), DEV)
    beff, E1, E2 = dev(t(G['beff'])), dev(t(G['E1'])), dev(t(G['E2']))
    A, B = beffective.beff2ab(beff, E1=E1, E2=E2, γ=c['γ'], dt=c['dt'])
    assert A.shape == (1, 3, 3, 3) and B.shape == (1, 3, 3)
    assert_close(A, G['A'], tag, 'A')
    assert_close(B, G['B'], tag, 'B')
    Mo = slowsims.blochsim_ab(c['M0'], A, B)
    assert_close(Mo, G['Mo'], tag, 'Mo3')
    if tag == 'f64':
        assert max_abs(Mo, MO0_RELAX) <= 1e-9
    A0, B0 = beffective.beff2ab(beff, γ=c['γ'], dt=c['dt'])          # defaults E1 = E2 = 0
    assert_close(A0, G['A_E0'], tag, 'A (E = 0)')
    assert_close(B0, G['B_E0'], tag, 'B (E = 0)')
    # gradient chain to rf, gr (through the differentiable composition)
    rf, gr = c['rf'].clone().requires_grad_(True), c['gr'].clone().requires_grad_(True)
    b = beffective.rfgr2beff(rf, gr, c['loc'], Δf=c['Δf'], b1Map=c['b1Map'], γ=c['γ'])
    Ag, Bg = beffective.beff2ab(b, E1=E1, E2=E2, γ=c['γ'], dt=c['dt'])
    # same numbers from the one-kernel and the differentiable route, bit for bit
    A1, B1 = beffective.beff2ab(b.detach(), E1=E1, E2=E2, γ=c['γ'], dt=c['dt'])
    assert torch.equal(Ag.detach(), A1) and torch.equal(Bg.detach(), B1)
    slowsims.blochsim_ab(c['M0'], Ag, Bg).sum().backward()
    assert_close(rf.grad, G['grad_rf'], tag, 'grad_rf through A, B')
    assert_close(gr.grad, G['grad_gr'], tag, 'grad_gr through A, B')
    # blochsim_ab's own gradients
    M = c['M0'].clone().requires_grad_(True)
    Ad, Bd = dev(t(G['A'])).requires_grad_(True), dev(t(G['B'])).requires_grad_(True)
    w = ((torch.arange(9, dtype=torch.float64) * 5) % 7 - 3).reshape(1, 3, 3).to(DT[tag])
    (slowsims.blochsim_ab(M, Ad, Bd) * dev(w)).sum().backward()
    assert_close(M.grad, G['ab_gM'], tag, 'blochsim_ab gM')
    assert_close(Ad.grad, G['ab_gA'], tag, 'blochsim_ab gA')
    assert_close(Bd.grad, G['ab_gB'], tag, 'blochsim_ab gB')


@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_ab_line_and_shapes(tag):
    r"""512-spin line with per-spin E1/E2 vs the reference's output; odd shapes (N = 2, 2-D Nd,
    nT not a multiple of the chunk, unaligned views) vs the oracle; and the defining property
    A M + B == blochsim(M) at 32^3 x 512."""
    G5, c5 = golden(f'ab512_{tag}'), to_dev(cases.ref_case(512, DT[tag], seed=1234), DEV)
    b5 = beffective.rfgr2beff(c5['rf'], c5['gr'], c5['loc'], Δf=c5['Δf'], b1Map=c5['b1Map'], γ=c5['γ'])
    A5, B5 = beffective.beff2ab(b5, E1=dev(t(G5['E1'])), E2=dev(t(G5['E2'])), γ=c5['γ'], dt=c5['dt'])
    assert_close(A5, G5['A'], tag, 'A 512')
    assert_close(B5, G5['B'], tag, 'B 512')
    assert_close(slowsims.blochsim_ab(c5['M0'], A5, B5), G5['Mo'], tag, 'Mo 512')
    g = torch.Generator().manual_seed(23)
    for shape, nT in (((2, 5, 7), 37), ((1, 70), 16), ((3, 1), 1), ((1, 0), 8), ((1, 4), 0)):
        beff = (torch.randn(shape + (nT + 1, 3), generator=g, dtype=torch.float64) * 0.5).to(DT[tag])
        beff = beff[..., 1:, :]                                   # unaligned, non-contiguous view
        E1 = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float64)).to(DT[tag])
        E2 = (0.8 + 0.2 * torch.rand(shape[:1] + (1,) * (len(shape) - 1), generator=g,
                                     dtype=torch.float64)).to(DT[tag])
        γ, dt = torch.tensor(4257.6, dtype=DT[tag]), torch.tensor(4e-6, dtype=DT[tag])
        Ao, Bo = O.beff2ab(beff, E1=E1, E2=E2, γ=γ, dt=dt)
        Ah, Bh = beffective.beff2ab(dev(beff), E1=dev(E1), E2=dev(E2), γ=dev(γ), dt=dev(dt))
        assert Ah.shape == Ao.shape and Bh.shape == Bo.shape
        assert_close(Ah, Ao, tag, f'A {shape} x {nT}')
        assert_close(Bh, Bo, tag, f'B {shape} x {nT}')
    # A M + B == stepping M (the fused 4-column kernel shares K1's arithmetic)
    sp, p = synth.cube_spins(32, dtype=DT[tag], device=DEV, seed_M0=3), synth.pulse(512, dtype=DT[tag], device=DEV)
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    E1, E2 = torch.exp(-p['dt'] / sp['T1']), torch.exp(-p['dt'] / sp['T2'])
    A, B = beffective.beff2ab(beff, E1=E1, E2=E2, γ=sp['γ'], dt=p['dt'])
    g2 = 2 * np.pi * sp['γ'] * p['dt']
    want = sims.blochsim_consts(sp['M0'], beff, γ2πdt=g2, E1=E1, E1_1=E1 - 1, E2=E2)
    assert_close(slowsims.blochsim_ab(sp['M0'], A, B), want, tag, 'A M + B vs blochsim, 32^3 x 512')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        beffective.beff2ab(beff.cpu())


@pytest.mark.usefixtures('host_constants')
def test_interp_grid_cache_sees_new_dwell_time():
    r"""interpT caches its grid per (dt, dt_new) tensors: an in-place change of either gives a
    new grid (different sample count), equal dwell times pass the inputs through."""
    from mrphy_amd import interp
    rf, gr = dev(torch.rand(1, 2, 64)), dev(torch.rand(1, 3, 64))
    dt, dt_new = dev(torch.tensor([8e-6])), dev(torch.tensor([4e-6]))
    a = interp.interpT(rf, gr, dt, dt_new)
    b = interp.interpT(rf, gr, dt, dt_new)                 # cached grid
    assert a[0].shape[2] == 128 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    dt_new.mul_(0.5)                                       # 2e-6: four times as many samples
    c = interp.interpT(rf, gr, dt, dt_new)
    assert c[0].shape[2] == 256 and float(c[2]) == float(dt_new)
    dt.copy_(dt_new)
    d = interp.interpT(rf, gr, dt, dt_new)
    assert d[0] is rf and d[1] is gr


# ---------------------------------------------------------------------------------------------
# install(): device-resident objects
# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
def test_device_object_glue_identity_keyed_mask_index():
    r"""What ``install()`` hangs on ``mobjs.SpinArray.extract/embed`` and ``SpinCube._update_loc_``
    for device-resident objects, called the way mobjs calls them (``mobjs.py:427-433,449,815-839``),
    TWICE per mask: the second lookup of the per-mask index used to compare mask tensors with ==.
    (The reference itself cannot travel to the GPU box; the objects here are stand-ins with the
    attributes those methods read.)"""
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand((1, 5, 6, 7), generator=g) > 0.4).to(DEV)
    arr = types.SimpleNamespace(device=DEV, mask=mask)
    nM = int(mask.sum())
    v = torch.randn((2, 5, 6, 7, 3), generator=g).to(DEV)
    for _ in range(3):
        v_ = mrphy_amd._spinarray_extract(arr, v)
        assert torch.equal(v_, v[mask.expand(2, -1, -1, -1)].reshape(2, nM, 3))
        back = mrphy_amd._spinarray_embed(arr, v_)
        assert torch.equal(mrphy_amd._spinarray_extract(arr, back), v_)
        assert torch.isnan(back[~mask.expand(2, -1, -1, -1)]).all()
    assert len(mrphy_amd._mask_index) >= 1
    ix = mrphy_amd._index_of(mask)
    assert mrphy_amd._index_of(mask) is ix                     # cached by identity
    other = mask.clone()
    assert mrphy_amd._index_of(other) is not ix                # equal values, different tensor
    fov, ofst = torch.tensor([[24., 24., 24.]], device=DEV), torch.tensor([[0., 1., -2.]], device=DEV)
    cube = types.SimpleNamespace(spinarray=arr, fov=fov, ofst=ofst,
                                 loc_=torch.empty((1, nM, 3), device=DEV))
    mrphy_amd._spincube_update_loc_(cube)
    mrphy_amd._spincube_update_loc_(cube)
    want = O.cube_loc(mask.cpu(), fov.cpu(), ofst.cpu())
    assert torch.equal(cube.loc_.cpu(), want)


# ---------------------------------------------------------------------------------------------
# interpT: the one-tap kinds
# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('kind', ['nearest', 'nearest-up', 'previous', 'next', 'zero'])
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_interpT_select_kinds_vs_scipy(kind, tag):
    r"""``interpT(kind=...)`` on the device against ``scipy.interpolate.interp1d`` applied the way
    ``Pulse.interpT`` applies it (``mobjs.py:201-215``: zero sample prepended, time axis 2): forward
    BIT-IDENTICAL (a selection has no arithmetic), single- and multi-coil rf, up- and down-sampling;
    the adjoint against the dense transpose of the selection."""
    import numpy as np
    from scipy import interpolate
    from mrphy_amd import interp
    dt_ = DT[tag]
    g = torch.Generator().manual_seed(31)
    for nT, dt_o, dt_n, nC in ((96, 8e-6, 4e-6, None), (130, 4e-6, 1.3e-5, 3), (64, 4e-6, 4e-6 * 0.37, None)):
        rf = torch.randn((2, 2, nT) + ((nC,) if nC else ()), generator=g, dtype=torch.float64).to(dt_)
        gr = torch.randn((2, 3, nT), generator=g, dtype=torch.float64).to(dt_)
        dt, dtn = torch.tensor([dt_o], dtype=torch.float64), torch.tensor([dt_n], dtype=torch.float64)
        t_o = np.arange(0, nT + 1) * dt.item()
        t_n = np.arange(1, t_o[-1] // dtn.item() + 1) * dtn.item()

        def ref(x):                                    # mobjs.py:203-215
            x0 = np.concatenate([np.zeros_like(x[:, :, :1]), x], axis=2)
            return interpolate.interp1d(t_o, x0, axis=2, kind=kind, copy=False, assume_sorted=True)(t_n)
        rf_h, gr_h = _leaf(rf, DEV), _leaf(gr, DEV)
        rf_n, gr_n, dt_out = interp.interpT(rf_h, gr_h, dev(dt), dev(dtn), kind=kind)
        assert rf_n.dtype == dt_ and rf_n.shape[2] == len(t_n) == gr_n.shape[2]
        assert rf_n.shape == ref(rf.numpy()).shape
        assert np.array_equal(rf_n.detach().cpu().numpy(), ref(rf.numpy()).astype(rf.numpy().dtype))
        assert np.array_equal(gr_n.detach().cpu().numpy(), ref(gr.numpy()).astype(gr.numpy().dtype))
        assert float(dt_out) == float(dtn.to(dt_))
        # adjoint: d/dy sum(w * select(y)) = S^T w, with S the 0/1 selection matrix (prepended column dropped)
        sel, nTn = interp.interp_select(nT, dt.item(), dtn.item(), kind)
        S = torch.zeros((nTn, nT + 1), dtype=torch.float64)
        S[torch.arange(nTn), torch.from_numpy(sel).long()] = 1
        S = S[:, 1:]
        w_rf = torch.randn(rf_n.shape, generator=g, dtype=torch.float64).to(dt_)
        w_gr = torch.randn(gr_n.shape, generator=g, dtype=torch.float64).to(dt_)
        ((rf_n * dev(w_rf)).sum() + (gr_n * dev(w_gr)).sum()).backward()
        want_gr = torch.einsum('jt,ncj->nct', S, w_gr.double())
        want_rf = (torch.einsum('jt,ncjk->nctk', S, w_rf.double()) if nC
                   else torch.einsum('jt,ncj->nct', S, w_rf.double()))
        assert gr_h.grad.shape == gr.shape and rf_h.grad.shape == rf.shape
        assert_close(gr_h.grad, want_gr, tag, f'{kind} d/dgr')
        assert_close(rf_h.grad, want_rf, tag, f'{kind} d/drf')
    # equal dwell times pass the inputs through, as for 'linear'
    same = interp.interpT(rf_h, gr_h, dev(dt), dev(dt.clone()), kind=kind)
    assert same[0] is rf_h and same[1] is gr_h


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


@pytest.mark.parametrize('dtype, tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('shapes', ['per_spin', 'scalars', 'batch'])
def test_slowsims_freeprec_gradients_wrt_dur_T1_T2_df(dtype, tol, shapes):
    r"""``slowsims.freeprec`` differentiated w.r.t. ``M, dur, T1, T2, Δf`` against autograd through the oracle's
    restatement of the reference's plain torch ops (``slowsims.py:151-174``): per-spin maps, 0-dim scalars, and a
    batch of two with ``dur (N,)`` -- the broadcast shapes the reference accepts."""
    import bloch_oracle as O
    g = torch.Generator(device='cpu').manual_seed(11)
    N, nM = (2, 37) if shapes == 'batch' else (1, 130)
    f64 = torch.float64
    M = torch.rand((N, nM, 3), generator=g, dtype=f64)
    if shapes == 'scalars':
        ops = dict(dur=torch.tensor(3e-3, dtype=f64), T1=torch.tensor(1.1, dtype=f64), T2=torch.tensor(0.07, dtype=f64),
                   Δf=torch.rand((1, 1), generator=g, dtype=f64) * 80 - 40)
    else:
        ops = dict(dur=(torch.rand((N,), generator=g, dtype=f64) + 0.5) * 4e-3,
                   T1=torch.rand((N, nM), generator=g, dtype=f64) + 0.6,
                   T2=torch.rand((N if shapes == 'batch' else 1, 1 if shapes == 'batch' else nM), generator=g, dtype=f64) * 0.1 + 0.03,
                   Δf=torch.rand((N, nM), generator=g, dtype=f64) * 200 - 100)
    w = torch.rand((N, nM, 3), generator=g, dtype=f64)
    names = ('dur', 'T1', 'T2', 'Δf')
    ref = _leafs(ops, names, f64)
    Mr = M.clone().requires_grad_(True)
    (O.freeprec_slow(Mr, ref['dur'], T1=ref['T1'], T2=ref['T2'], Δf=ref['Δf']) * w).sum().backward()
    got = _leafs({k: v.to(DEV) for k, v in ops.items()}, names, dtype)
    Mg = M.to(DEV, dtype).requires_grad_(True)
    from mrphy_amd import slowsims
    out = slowsims.freeprec(Mg, got['dur'], T1=got['T1'], T2=got['T2'], Δf=got['Δf'])
    (out * w.to(DEV, dtype)).sum().backward()
    pairs = [('M', Mg.grad, Mr.grad)] + [(k, got[k].grad, ref[k].grad) for k in names]
    for k, a_, b_ in pairs:
        assert a_ is not None and a_.shape == b_.shape, k
        d = float((a_.double().cpu() - b_).norm() / b_.norm())
        assert d <= tol, (k, d)
    # no relaxation / no precession: the absent operands are simply absent
    d2 = got['dur'].detach().clone().requires_grad_(True)
    slowsims.freeprec(Mg.detach(), d2, Δf=got['Δf'].detach()).sum().backward()
    d3 = ops['dur'].clone().requires_grad_(True)
    O.freeprec_slow(M, d3, Δf=ops['Δf']).sum().backward()
    assert float((d2.grad.double().cpu() - d3.grad).norm() / d3.grad.norm()) <= tol


@pytest.mark.parametrize('dtype, tol', [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_beff2ab_gradients_wrt_E1_E2_gamma_dt(dtype, tol):
    r"""``beff2ab`` differentiated w.r.t. ``beff, E1, E2, γ, dt`` against autograd through the oracle's restatement of
    the reference's loop (``beffective.py:73-100``): per-spin ``E1, E2``, 0-dim ``γ``, ``dt (N,)``; a ``γ = 0``
    spin in the batch (its γ2πdt is zero: the round-3 form divided by it)."""
    import bloch_oracle as O
    g = torch.Generator(device='cpu').manual_seed(12)
    f64 = torch.float64
    N, nM, nT = 2, 70, 24
    beff = torch.randn((N, nM, nT, 3), generator=g, dtype=f64) * 0.4
    ops = dict(E1=1 - torch.rand((N, nM), generator=g, dtype=f64) * 1e-2, E2=1 - torch.rand((1, nM), generator=g, dtype=f64) * 5e-2,
               γ=torch.full((N, nM), 4257.6, dtype=f64), dt=torch.tensor([4e-6, 6e-6], dtype=f64))
    ops['γ'][0, 3] = 0.0
    wA, wB = torch.rand((N, nM, 3, 3), generator=g, dtype=f64), torch.rand((N, nM, 3), generator=g, dtype=f64)
    names = ('E1', 'E2', 'γ', 'dt')
    ref = _leafs(ops, names, f64)
    br = beff.clone().requires_grad_(True)
    A, B = O.beff2ab(br, **ref)
    ((A * wA).sum() + (B * wB).sum()).backward()
    got = _leafs({k: v.to(DEV) for k, v in ops.items()}, names, dtype)
    bg = beff.to(DEV, dtype).requires_grad_(True)
    A2, B2 = beffective.beff2ab(bg, **got)
    ((A2 * wA.to(DEV, dtype)).sum() + (B2 * wB.to(DEV, dtype)).sum()).backward()
    for k, a_, b_ in [('beff', bg.grad, br.grad)] + [(k, got[k].grad, ref[k].grad) for k in names]:
        assert a_ is not None and a_.shape == b_.shape and bool(torch.isfinite(a_).all()), k
        d = float((a_.double().cpu() - b_).norm() / b_.norm())
        assert d <= tol, (k, d)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_beff2ab_of_an_empty_pulse_has_exact_zero_constant_gradients(dtype):
    r"""nT = 0: ``A = I``, ``B = 0`` whatever the constants, so ``dL/d{E1, E2, γ, dt}`` are exact zeros -- as
    the reference's autograd gives them (``beffective.py:73-100``) -- not whatever the gradient buffer held."""
    N, nM = 2, 70
    beff = torch.zeros((N, nM, 0, 3), dtype=dtype, device=DEV, requires_grad=True)
    ops = dict(E1=torch.full((N, nM), 0.99, dtype=dtype, device=DEV), E2=torch.full((1, nM), 0.9, dtype=dtype, device=DEV),
               γ=torch.full((N, nM), 4257.6, dtype=dtype, device=DEV), dt=torch.tensor([4e-6, 6e-6], dtype=dtype, device=DEV))
    got = _leafs(ops, ('E1', 'E2', 'γ', 'dt'), dtype)
    junk = torch.full((N * nM * 4 + 64,), float('nan'), dtype=dtype, device=DEV)     # what a fresh allocation may hold
    del junk
    A, B = beffective.beff2ab(beff, **got)
    assert torch.equal(A, torch.eye(3, dtype=dtype, device=DEV).expand(N, nM, 3, 3)) and not bool(B.any())
    (A.sum() + B.sum()).backward()
    for k, v in got.items():
        assert v.grad is not None and v.grad.shape == v.shape and not bool(v.grad.any()), k
