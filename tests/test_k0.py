r"""K0: ``beffective.rfgr2beff`` and its adjoint -- golden variants, the reference's own fp32 ``Beff`` rows bit for bit, every
transmit-coil path, the ``out=`` / ``store=`` extensions and the placement-aware arena.

Regrouped by component in round 5 from ``test_hip_parity.py`` / ``test_hip_round{2,3,4}.py`` (no assertion changed; each test keeps its name).
"""
import pytest

from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_rfgr2beff_variants(tag):
    G = golden(f'rfgr_{tag}')
    for name, kw in cases.rfgr_variants(DT[tag]).items():
        kw = to_dev(kw, DEV)
        rf, gr, loc = kw.pop('rf'), kw.pop('gr'), kw.pop('loc')
        rf, gr = rf.clone().requires_grad_(True), gr.clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, loc, **kw)
        assert beff.is_contiguous() and beff.shape == G[f'{name}.beff'].shape
        assert_close(beff, G[f'{name}.beff'], tag, f'{name}.beff')
        w = torch.cos(torch.arange(beff.numel(), dtype=torch.float64) * 0.37).reshape(beff.shape)
        (beff * w.to(device=DEV, dtype=DT[tag])).sum().backward()
        assert rf.grad.shape == rf.shape and gr.grad.shape == gr.shape
        assert_close(rf.grad, G[f'{name}.grad_rf'], tag, f'{name}.grad_rf')
        assert_close(gr.grad, G[f'{name}.grad_gr'], tag, f'{name}.grad_gr')


@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_rfgr2beff_map_gradients(tag):
    r"""loc / Δf / b1Map / γ gradients (the reference gets them from autograd)."""
    v = cases.rfgr_variants(DT[tag])['ptx4']
    names = ('loc', 'Δf', 'b1Map', 'γ')
    ref = {k: v[k].clone().requires_grad_(True) for k in names}
    b = O.rfgr2beff(v['rf'], v['gr'], ref['loc'], Δf=ref['Δf'], b1Map=ref['b1Map'], γ=ref['γ'])
    w = torch.cos(torch.arange(b.numel(), dtype=torch.float64) * 0.37).reshape(b.shape).to(DT[tag])
    (b * w).sum().backward()
    hip = {k: v[k].to(DEV).requires_grad_(True) for k in names}
    bh = beffective.rfgr2beff(dev(v['rf']), dev(v['gr']), hip['loc'], Δf=hip['Δf'],
                              b1Map=hip['b1Map'], γ=hip['γ'])
    (bh * w.to(DEV)).sum().backward()
    for k in names:
        assert_close(hip[k].grad, ref[k].grad, tag, f'grad {k}')


@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('tag', ['f64', 'f32'])
@pytest.mark.parametrize('nC,nT', [(2, 32), (3, 32), (4, 33), (8, 37), (9, 32), (12, 600), (16, 37), (17, 32), (32, 37), (33, 32),
                                   (40, 37), (41, 32), (48, 32), (64, 37), (65, 32), (70, 24)])
def test_coil_count_paths(tag, nC, nT):
    r"""Every coil-count branch of K0 and K2: the register/LDS builds hold up to 8, 16 or 32 coils
    (2, 3, 8 | 9, 16 | 17, 32: partly and completely filled); round 4: fp32 K0 and K2 go on to capacities 40 / 48 /
    64 (33, 40 | 41, 48 | 64) and the K0 adjoint walks any coil count in blocks of 32 (33, 64, 65, 70: one, two and
    three blocks, the last one partly filled); beyond 64 coils -- and beyond 32 in fp64 -- the generic forward
    kernels run; the exact
    counts 4, 8, 12, 16 take K0's packed-scalar kernel (two time points per thread: odd pulse lengths leave
    a half-filled thread at the row end; 600 steps span two time tiles);
    the fused adjoint covers 2-8 coils, beyond that the composed one runs.  nT = 37 leaves a tail of
    5 steps after the 8-step chunks (the strided staging of the tail's rf samples).  Forward and
    gradients vs the oracle; fused forward == rfgr2beff + blochsim bit for bit at every count: the
    coil sum is one ascending FMA chain in every build."""
    dt_ = DT[tag]
    gen = torch.Generator().manual_seed(100 + nC)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64)  # noqa: E731
    N, nM = 2, 70
    M0 = rnd(N, nM, 3).to(dt_)
    rf, gr = ((rnd(N, 2, nT, nC) * 2 - 1) * 1.5).to(dt_), (rnd(N, 3, nT) * 2 - 1).to(dt_)
    loc, df = ((rnd(N, nM, 3) * 2 - 1) * 6).to(dt_), ((rnd(N, nM) * 2 - 1) * 200).to(dt_)
    b1 = ((rnd(N, nM, 2, nC) * 2 - 1) * 0.7).to(dt_)
    T1, T2 = (0.5 + rnd(N, nM)).to(dt_), (0.02 + 0.1 * rnd(N, nM)).to(dt_)
    γ, dt = torch.tensor(4257.6, dtype=dt_), torch.tensor([4e-6], dtype=dt_)

    def run(kind):
        on = (lambda x: x) if kind == 'oracle' else dev
        r, g = on(rf).clone().requires_grad_(True), on(gr).clone().requires_grad_(True)
        kw = dict(T1=on(T1), T2=on(T2), γ=on(γ), dt=on(dt))
        if kind == 'oracle':
            be = O.rfgr2beff(r, g, loc, Δf=df, b1Map=b1, γ=γ)
            Mo = O.blochsim(M0, be, **kw)
        elif kind == 'two':
            be = beffective.rfgr2beff(r, g, dev(loc), Δf=dev(df), b1Map=dev(b1), γ=dev(γ))
            Mo = sims.blochsim(dev(M0), be, **kw)
        else:
            be = None
            Mo = fused.blochsim_rfgr(dev(M0), r, g, dev(loc), Δf=dev(df), b1Map=dev(b1), γ_beff=dev(γ), **kw)
        Mo.sum().backward()
        return (None if be is None else be.detach()), Mo.detach(), r.grad, g.grad
    ora, two, fu = run('oracle'), run('two'), run('fused')
    assert_close(two[0], ora[0], tag, 'beff')
    assert max_abs(fu[1], two[1]) == 0.0
    for i, nm in ((1, 'Mo'), (2, 'grad_rf'), (3, 'grad_gr')):
        assert_close(two[i], ora[i], tag, f'two-kernel {nm}')
        assert_close(fu[i], ora[i], tag, f'fused {nm}')


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


def test_rfgr2beff_out_is_the_same_bits_and_the_same_gradients():
    sp, p, kw = _problem(10, 96)
    fresh = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    blk = torch.full_like(fresh, float('nan'))
    got = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
    assert got.data_ptr() == blk.data_ptr() and torch.equal(got, fresh)
    # differentiable through the caller's block as well
    g = []
    for out in (None, torch.empty_like(fresh)):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        b = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=out)
        sims.blochsim(sp['M0'], b, **kw).sum().backward()
        g.append((rf.grad, gr.grad))
    assert torch.equal(g[0][0], g[1][0]) and torch.equal(g[0][1], g[1][1])
    with pytest.raises(AssertionError):                       # wrong shape
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], out=torch.empty((1, 5, 96, 3), device=DEV))
    with pytest.raises(AssertionError):                       # wrong dtype
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], out=blk.double())
    with pytest.raises(AssertionError):                       # nothing to write lazily
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], lazy=True, out=blk)


def test_beff_arena_probes_and_keeps_one_block():
    sp, p, kw = _problem(12, 64)
    calls = []

    def probe(b):
        calls.append(b.data_ptr())
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=b)
        sims.blochsim(sp['M0'], b, **kw)
    shape = (1, 12 ** 3, 64, 3)
    arena = workspace.BeffArena(shape, torch.float32, DEV, probe, candidates=3, reps=2)
    rep = arena.report
    assert tuple(arena.block.shape) == shape and len(rep['candidate_ms']) == 3 and len(set(rep['ptr'])) == 3
    assert arena.block.data_ptr() == int(rep['ptr'][rep['chosen']], 16)
    assert rep['candidate_ms'][rep['chosen']] == min(rep['candidate_ms'])
    assert len(calls) == 3 * 4                                 # first touch + one untimed + two timed launches per block
    Mo = sims.blochsim(sp['M0'], beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'],
                                                      out=arena.block), **kw)
    ref = sims.blochsim(sp['M0'], beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']), **kw)
    assert torch.equal(Mo, ref)
    one = workspace.BeffArena(shape, torch.float32, DEV, None)  # no probe: one block, nothing timed
    assert one.report['candidate_ms'] == [] and one.report['chosen'] == 0
    with pytest.raises(ValueError):
        workspace.BeffArena(shape, torch.float32, torch.device('cpu'), probe)


@pytest.mark.parametrize('cfg', [1, 2, 4])
def test_k0_rows_equal_the_reference_beff(cfg):
    r"""the all-spins 1e-5 assertions compare with an exact integration of a field that the oracle's C
    restatement forms in single precision "as the reference forms its fp32 Beff".  This pins that premise to the
    reference itself: ``tests/golden/big_beff_rows_f32.npz`` holds the reference's own ``rfgr2beff`` output (fp32,
    CPU) for eight spins of each BASELINE config (config 4: on the reference's ``interpT`` pulse); K0 must return
    those rows BIT FOR BIT, and the oracle's single-precision field must equal them too."""
    import numpy as np
    import cases
    import bloch_c as C
    with np.load(os.path.join(ROOT, 'tests', 'golden', 'big_beff_rows_f32.npz'), allow_pickle=False) as z:
        idx, want = torch.from_numpy(z[f'cfg{cfg}.idx']), torch.from_numpy(z[f'cfg{cfg}.beff'])
    idx_all, sp, pulse = cases.big_subset(cfg, torch.float32, 4096)
    assert torch.equal(idx_all[:idx.numel()], idx)
    if cfg == 4:                                              # the reference's interpT output, from its own fixture
        from util import golden
        Ig = golden('interp_f32')
        pulse = dict(rf=torch.from_numpy(Ig['rf']), gr=torch.from_numpy(Ig['gr']), dt=torch.from_numpy(Ig['dt']))
    sl = slice(0, idx.numel())
    got = beffective.rfgr2beff(pulse['rf'].to(DEV), pulse['gr'].to(DEV), sp['loc'][:, sl].to(DEV),
                               Δf=sp['Δf'][:, sl].to(DEV), γ=sp['γ'].to(DEV))
    assert got.shape == want.shape
    assert torch.equal(got.cpu(), want), float((got.cpu().double() - want.double()).abs().max())
    if hasattr(C, 'field_f32'):
        f = C.field_f32(pulse['rf'], pulse['gr'], sp['loc'][:, sl], Δf=sp['Δf'][:, sl], γ_beff=sp['γ'])
        assert torch.equal(f, want)


@pytest.mark.parametrize('nC', [2, 9, 33, 64, 65, 100, 130])
def test_k0_any_coil_count_is_the_oracles_fp32_field_bit_for_bit(nC):
    r"""``rfgr2beff`` with a b1 map at any coil count -- register capacities 8...64, and beyond 64 coils in blocks of 64 whose
    launches continue the ascending FMA chains from the stored values (round 4: 65 coils used to take the generic kernel) --
    equals oracle/bloch_c.c's single-precision field (the reference's own fp32 rows, ``beffective.py:153-165``) bit for bit;
    pulse lengths with and without a ragged last thread, a spin count that is not a multiple of the block's rows."""
    import bloch_c as C
    g = torch.Generator().manual_seed(900 + nC)
    rnd = lambda *s: (torch.rand(s, generator=g, dtype=torch.float64) * 2 - 1).float()  # noqa: E731
    for nT, nM in ((37, 70), (64, 131)):
        rf, gr, loc = rnd(1, 2, nT, nC) * 0.3, rnd(1, 3, nT) * 2, rnd(1, nM, 3) * 8
        b1, df = rnd(1, nM, 2, nC), rnd(1, nM) * 300
        γ = torch.tensor(4257.6, dtype=torch.float32)
        want = C.field_f32(rf, gr, loc, Δf=df, b1Map=b1, γ_beff=γ.double())
        d = lambda x: x.to(DEV)  # noqa: E731
        got = beffective.rfgr2beff(d(rf), d(gr), d(loc), Δf=d(df), b1Map=d(b1), γ=d(γ))
        assert torch.equal(got.cpu(), want), f'nC={nC} nT={nT} nM={nM}: {float((got.cpu() - want).abs().max()):.3e}'
        blk = torch.full_like(got, float('nan'))                  # into a caller-owned block as well
        assert torch.equal(beffective.rfgr2beff(d(rf), d(gr), d(loc), Δf=d(df), b1Map=d(b1), γ=d(γ), out=blk), got)


def test_multicoil_k0_is_not_an_order_of_magnitude_off_the_one_coil_kernel():
    r"""A coarse guard, not a benchmark: the parallel-transmit K0 builds write the same bytes as the one-coil kernel, and for a
    few commits of round 4 they were 6 x slower (write-through `sc1 nt` on their 4- / 8-byte stores).  8 and 33 coils must stay
    within 4 x / 8 x of one coil at 48^3 x 512 (measured: 0.16, 0.24 and 0.53 ms; 33 coils is compute-bound)."""
    import statistics
    sp, p, _ = _problem(48, 512)
    g = torch.Generator().manual_seed(5)

    def t_k0(nC):
        rf = p['rf'] if nC == 1 else (torch.rand((1, 2, 512, nC), generator=g) * 0.02).to(DEV)
        b1 = None if nC == 1 else torch.rand((1, 48 ** 3, 2, nC), generator=g).to(DEV)
        ts = []
        with torch.no_grad():
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        return statistics.median(ts[2:])
    t1, t8, t33 = t_k0(1), t_k0(8), t_k0(33)
    print(f'K0 48^3 x 512: 1 coil {t1:.3f} ms, 8 coils {t8:.3f} ms, 33 coils {t33:.3f} ms')
    assert t8 <= 4 * t1 and t33 <= 8 * t1, (t1, t8, t33)


def test_store_policy_never_changes_the_bits_and_the_arena_reports_one():
    r"""``rfgr2beff(..., store=)`` (ABI 4: ``mrphy_rfgr2beff_st``): one coil (16-byte stores) and five coils (12-byte threads)
    give the same bits under every cache policy of the stores; an unknown policy raises; a two-argument probe makes the
    arena time both policies per block and report the one it kept."""
    from mrphy_amd import _lib
    sp, p, kw = _problem(12, 96)
    g = torch.Generator().manual_seed(8)
    rf5 = (torch.rand((1, 2, 96, 5), generator=g) * 0.02).to(DEV)
    b15 = torch.rand((1, 12 ** 3, 2, 5), generator=g).to(DEV)
    for rf, b1 in ((p['rf'], None), (rf5, b15)):
        ref = beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        for store in ('auto', 'plain', 'nt', 'sc1nt'):
            got = beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'], store=store)
            assert torch.equal(got, ref), store
    with pytest.raises(AssertionError):
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], store='streaming')
    lib = _lib.require_library()                                   # the C entry point refuses what the header does not name
    assert lib.mrphy_rfgr2beff_st(0, None, 0, None, 0, None, None, 0, 0, None, 0, 0, None, None, 1, 1, 1, 1, 3, None) != 0
    seen = []

    def probe(b, store):
        seen.append(store)
        beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=b, store=store)
        sims.blochsim(sp['M0'], b, **kw)
    arena = workspace.BeffArena((1, 12 ** 3, 96, 3), torch.float32, DEV, probe, candidates=2, reps=2)
    rep = arena.report
    assert arena.store in ('sc1nt', 'nt') and rep['store'] == arena.store
    assert set(rep['by_store']) == {'sc1nt', 'nt'} and all(len(v) == 2 for v in rep['by_store'].values())
    assert seen.count('nt') == 2 * 3 and seen.count('sc1nt') == 2 * 4   # per block and policy: one untimed + two timed launches
    #                                                                    (+ the block's first touch, under the first policy)
    assert rep['candidate_ms'][rep['chosen']] == min(rep['candidate_ms'])
    Mo = sims.blochsim(sp['M0'], beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=arena.block,
                                                      store=arena.store), **kw)
    assert torch.equal(Mo, sims.blochsim(sp['M0'], beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']), **kw))
