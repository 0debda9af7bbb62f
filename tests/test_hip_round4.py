r"""Round-4 additions to the GPU suite (``-m gpu``, through the C ABI):

* ``rfgr2beff(..., out=)`` and the placement-aware ``workspace.BeffArena``: same bits as a fresh tensor, same
  gradients, misuse raises;
* K1's two tile orders (plain for the no-history kernel, XCD-contiguous for the history-saving one) and its two
  schedules (pinned 5-/6-step batches, unpinned 3-/4-step batches) are the same arithmetic: rows of a run on a
  whole grid equal a run on those rows alone, bit for bit -- on grids whose tile count is not a multiple of 8 as well;
* an fp32 fuzz (both precision modes' default, the precise step) against the fp64 oracle at the north star's 1e-5.
"""
import os

import pytest
import torch

import mrphy_amd
from mrphy_amd import beffective, sims, synth, workspace

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n, nT, dtype=torch.float32, seed=3, idx=None):
    sp = synth.cube_spins(n, idx, dtype=dtype, device=DEV, seed_M0=seed)
    p = synth.pulse(nT, dtype=dtype, device=DEV)
    return sp, p, dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])


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


@pytest.mark.parametrize('mode', ['precise', 'fast'])
@pytest.mark.parametrize('nM', [64 * 8 * 3, 64 * 13 + 5, 64 * 7])
def test_k1_xcd_tile_order_is_the_same_arithmetic(mode, nM):
    r"""The history-saving K1 walks the spin tiles in XCD-contiguous order (block b -> tile (b % 8) * per_xcd + b / 8,
    grid padded to a multiple of 8), the no-history K1 in plain order (round 4, second half: K0's `sc1 nt` stores made
    the XCD-contiguous order of the no-history kernel unnecessary; DESIGN.md §3 "K1 right behind K0").  Every row must
    be integrated exactly once and exactly as on its own in both -- tile counts that are multiples of 8, not
    multiples of 8 (blocks past the last tile exit), and fewer than 8."""
    n, nT = 16, 64                                             # nT % 32 == 0: the line-granular kernel
    idx = torch.arange(nM)
    sp, p, kw = _problem(n, nT, idx=idx)
    with mrphy_amd.precision(mode), torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        Mo = sims.blochsim(sp['M0'], beff, **kw)
        # the same rows alone (one tile each time: no tile order to speak of), and the fused kernel
        for lo in (0, 64 * 5, nM - 64):
            sl = slice(lo, lo + 64)
            part = sims.blochsim(sp['M0'][:, sl].contiguous(), beff[:, sl].contiguous(),
                                 T1=sp['T1'][:, sl], T2=sp['T2'][:, sl], γ=sp['γ'], dt=p['dt'])
            assert torch.equal(part, Mo[:, sl])
        from mrphy_amd import fused
        Mf = fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
        assert torch.equal(Mf, Mo)
    with mrphy_amd.precision(mode):
        Mh = sims.blochsim(sp['M0'].clone().requires_grad_(True), beff, **kw)       # history-saving twin
    assert torch.equal(Mh.detach(), Mo)


def _offset_copy(x, pad=2):
    r"""The same values at an address that is element-aligned but not 128-B-aligned: the launchers then take
    the chunked kernels instead of the line-granular ones."""
    buf = torch.empty(x.numel() + pad, dtype=x.dtype, device=x.device)
    y = buf[pad:].view(x.shape)
    y.copy_(x)
    assert y.data_ptr() % 128 != 0 and y.is_contiguous()
    return y


@pytest.mark.parametrize('relax', [True, False])
@pytest.mark.parametrize('nM', [64 * 9 + 7, 64 * 8])
def test_fp64_line_kernels_equal_the_chunked_ones(relax, nM):
    r"""Round 4: fp64 ``blochsim`` (forward, forward with history, adjoint) runs line-granular kernels when the
    rows sit on 128-B lines and nT % 16 == 0 (a line = 16 doubles, period 3 lines = 16 steps).  Same step
    arithmetic as the chunked kernels they replace there: outputs and gradients bit for bit -- ragged last tile,
    several 16-step periods, carries across all three piece boundaries."""
    f64 = torch.float64
    nT = 80                                                   # 5 periods
    sp, p, kw = _problem(16, nT, dtype=f64, idx=torch.arange(nM))
    if not relax:
        kw = dict(γ=kw['γ'], dt=kw['dt'])
    beff = beffective.rfgr2beff(p['rf'] * 40, p['gr'] * 3, sp['loc'], Δf=sp['Δf'], γ=sp['γ'])   # some steps beyond pi
    assert beff.data_ptr() % 128 == 0
    res = []
    for b in (beff, _offset_copy(beff)):
        b = b.detach().requires_grad_(True)
        Mi = sp['M0'].clone().requires_grad_(True)
        with torch.no_grad():
            Mo_ng = sims.blochsim(Mi, b, **kw)                # no history
        Mo = sims.blochsim(Mi, b, **kw)                       # with history
        gM, gB = torch.autograd.grad(Mo, (Mi, b), torch.cos(Mo.detach() * 3.0))
        res.append((Mo_ng, Mo.detach(), gM, gB))
    for a_, b_ in zip(*res):
        assert torch.equal(a_, b_)
    assert torch.equal(res[0][0], res[0][1])


def test_fp64_fused_with_many_coils_takes_the_composed_route():
    r"""fp64 with more than 8 transmit coils: no fused register build exists (it would spill); the host composes
    rfgr2beff + blochsim, the C ABI falls back to its generic build -- the same bits either way."""
    from mrphy_amd import fused, _lib, _host
    f64 = torch.float64
    nC, nT, n = 12, 32, 6
    sp, p, kw = _problem(n, nT, dtype=f64)
    g = torch.Generator(device='cpu').manual_seed(5)
    rf = torch.randn((1, 2, nT, nC), generator=g, dtype=f64).to(DEV) * 0.05
    b1 = torch.randn((1, n ** 3, 2, nC), generator=g, dtype=f64).to(DEV)
    with torch.no_grad():
        beff = beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
        want = sims.blochsim(sp['M0'], beff, **kw)
        got = fused.blochsim_rfgr(sp['M0'], rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ_beff=sp['γ'], **kw)
        assert torch.equal(got, want)
        # ... and the entry point itself (generic build behind the same C ABI)
        lib = _lib.require_library()
        ps = beffective._PulseOnSpins(rf, p['gr'], sp['loc'], sp['Δf'], b1, sp['γ'])
        γ2πdt, E1, E2, E1_1 = sims.relax_constants(sp['T1'], sp['T2'], sp['γ'], p['dt'], 4, DEV)
        code, gg, e1, e2, e1m1 = sims._prep_constants(γ2πdt, E1, E2, E1_1, ps.N, ps.Nd, f64, DEV)
        Mo = torch.empty_like(want)
        rc = lib.mrphy_blochsim_rfgr_fwd(code, sp['M0'].contiguous().data_ptr(), *ps.k0_args(), *gg.args, *e1.args,
                                         *e2.args, e1m1.t.data_ptr(), Mo.data_ptr(), None, 0, ps.N, ps.nM, ps.nT,
                                         ps.nC, _host.current_stream(DEV))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(Mo, want)


# ---------------------------------------------------------------------------------------------
# gradients w.r.t. the constants that the reference's autograd supplies (VERDICT r3 "missing" #3)
# ---------------------------------------------------------------------------------------------
def _leafs(d, names, dtype):
    return {k: (v.detach().clone().to(dtype).requires_grad_(True) if k in names and v is not None else v)
            for k, v in d.items()}


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


def test_blochsim_constant_gradients_with_a_gamma_zero_spin():
    r"""a spin with γ2πdt == 0 used to put 0/0 into the γ / dt gradient of ``slowsims.blochsim``; the
    adjoint now accumulates dL/db . B directly (no division)."""
    import bloch_oracle as O
    from mrphy_amd import slowsims
    g = torch.Generator(device='cpu').manual_seed(13)
    f64 = torch.float64
    N, nM, nT = 1, 66, 16
    M = torch.rand((N, nM, 3), generator=g, dtype=f64)
    beff = torch.randn((N, nM, nT, 3), generator=g, dtype=f64) * 0.3
    γ = torch.full((N, nM), 4257.6, dtype=f64); γ[0, 5] = 0.0
    ops = dict(T1=torch.rand((N, nM), generator=g, dtype=f64) + 0.5, T2=torch.rand((N, nM), generator=g, dtype=f64) * 0.1 + 0.03,
               γ=γ, dt=torch.tensor([4e-6], dtype=f64))
    ref = _leafs(ops, ('T1', 'T2', 'γ', 'dt'), f64)
    O.blochsim_slow(M, beff, **ref).sum().backward()
    got = _leafs({k: v.to(DEV) for k, v in ops.items()}, ('T1', 'T2', 'γ', 'dt'), f64)
    slowsims.blochsim(M.to(DEV), beff.to(DEV), **got).sum().backward()
    for k in ('T1', 'T2', 'γ', 'dt'):
        assert bool(torch.isfinite(got[k].grad).all()), k
        assert float((got[k].grad.cpu() - ref[k].grad).abs().max()) <= 1e-9 * max(1.0, float(ref[k].grad.abs().max())), k


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


def test_underflowed_relaxation_is_refused_by_the_precise_adjoint():
    r"""with ``E2 == 0`` (T2 < dt/100 in fp32) the precise adjoint would divide 0 by 0.  The host says so
    (once per constants: the check is cached) -- for the materialised and the fused route; the fast step and the
    forward alone are unaffected."""
    from mrphy_amd import fused
    sp, p, kw = _problem(6, 32)
    kw = dict(kw, T2=torch.full_like(sp['T2'], 1e-9))
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    with torch.no_grad():
        assert bool(torch.isfinite(sims.blochsim(sp['M0'], beff, **kw)).all())
    Mi = sp['M0'].clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match='exactly 0'):
        sims.blochsim(Mi, beff, **kw)
    rf = p['rf'].clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match='exactly 0'):
        fused.blochsim_rfgr(sp['M0'], rf, p['gr'], sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'], **kw)
    with mrphy_amd.precision('fast'):
        sims.blochsim(Mi, beff, **kw).sum().backward()
    assert bool(torch.isfinite(Mi.grad).all())


def test_fuzz_forward_fp32_vs_fp64_oracle():
    r"""60 random fp32 problems (the shapes of test_fuzz_forward_vs_c_restatement: batch 1-3, 1-200 spins, pulse lengths on and
    off the line grid up to 192 steps, 1-9 coils, with and without b1Map / Δf / relaxation): rfgr2beff + blochsim and the fused
    kernel agree bit for bit, and both are within the north star's 1e-5 (relative L2) of oracle/bloch_c.c run in fp64 on the same
    fp32 inputs -- field formed in fp64 too, so the bound includes the rounding of Beff to fp32."""
    import bloch_c as C
    from mrphy_amd import fused
    g = torch.Generator().manual_seed(int(os.environ.get('MRPHY_FUZZ_SEED', 20261005)))
    rnd = lambda *s: torch.rand(s, generator=g, dtype=torch.float64).float()  # noqa: E731
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    worst = 0.0
    for case in range(int(os.environ.get('MRPHY_FUZZ_CASES', 60))):
        N, nM = ri(1, 3), ri(1, 200)
        nT = (ri(1, 70), 32 * ri(1, 6), 16 * ri(1, 5))[ri(0, 2)]
        nC = (1, 1, 2, 5, 8, 9)[ri(0, 5)]
        Np = N if ri(0, 1) else 1
        has_b1 = bool(ri(0, 3))
        rf = ((rnd(Np, 2, nT, nC) if (nC > 1 or ri(0, 1)) else rnd(Np, 2, nT)) * 2 - 1) * 0.3
        gr = (rnd(Np, 3, nT) * 2 - 1) * 2
        loc = (rnd(N, nM, 3) * 2 - 1) * 8
        b1 = None
        if has_b1:
            b1 = (rnd(N, nM, 2, nC) * 2 - 1) if rf.ndim == 4 else (rnd(N, nM, 2) * 2 - 1)
        df = ((rnd(N, nM) * 2 - 1) * 300) if ri(0, 1) else None
        relax = bool(ri(0, 2))
        T1, T2 = 0.3 + rnd(N, nM), 0.01 + 0.1 * rnd(N, nM)
        γ, dt = torch.tensor(4257.6, dtype=torch.float32), torch.tensor([4e-6], dtype=torch.float32)
        M0 = rnd(N, nM, 3) * 2 - 1
        kw = dict(T1=T1, T2=T2) if relax else {}
        up = lambda x: None if x is None else x.double()  # noqa: E731
        want = C.blochsim_rfgr(up(M0), up(rf), up(gr), up(loc), Δf=up(df), b1Map=up(b1), γ_beff=up(γ), γ=up(γ), dt=up(dt),
                               **{k_: v.double() for k_, v in kw.items()})
        d = lambda x: None if x is None else x.to(DEV)  # noqa: E731
        kwd = {k_: v.to(DEV) for k_, v in kw.items()}
        beff = beffective.rfgr2beff(d(rf), d(gr), d(loc), Δf=d(df), b1Map=d(b1), γ=d(γ))
        two = sims.blochsim(d(M0), beff, γ=d(γ), dt=d(dt), **kwd)
        fu = fused.blochsim_rfgr(d(M0), d(rf), d(gr), d(loc), Δf=d(df), b1Map=d(b1), γ_beff=d(γ), γ=d(γ), dt=d(dt), **kwd)
        tag = f'case {case}: N={N} nM={nM} nT={nT} nC={nC} Np={Np} b1={has_b1} df={df is not None} relax={relax} rf.ndim={rf.ndim}'
        assert torch.equal(two, fu), tag
        rel = float((two.double().cpu() - want).norm() / want.norm())
        worst = max(worst, rel)
        assert rel <= 1e-5, tag + f' rel-L2 {rel:.3e}'
    print(f'fp32 fuzz: worst relative L2 {worst:.3e}')


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


def test_fuzz_gradients_fp32_vs_fp64_oracle():
    r"""30 random fp32 problems: gradients of a weighted sum of Mo w.r.t. Mi, rf, gr through rfgr2beff + blochsim and through the
    fused route against the torch oracle's autograd run in fp64 on the same fp32 inputs: relative L2 <= 3e-5 per gradient (the
    forward bound is 1e-5; a gradient sums nT x nM rounded contributions), the two routes' grad_Mi bit-identical when the fused
    adjoint runs (nT % 16 == 0, <= 8 coils)."""
    import bloch_oracle as O
    from mrphy_amd import fused
    g = torch.Generator().manual_seed(int(os.environ.get('MRPHY_FUZZ_SEED', 515151)))
    rnd = lambda *s: torch.rand(s, generator=g, dtype=torch.float64).float()  # noqa: E731
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    worst = {}
    for case in range(int(os.environ.get('MRPHY_FUZZ_CASES', 30))):
        N, nM = ri(1, 2), ri(8, 150)
        nT = (16, 32, 48, 96, ri(1, 60))[ri(0, 4)]
        nC = (1, 1, 1, 3, 8, 9)[ri(0, 5)]
        Np = N if ri(0, 1) else 1
        has_b1 = nC > 1 or bool(ri(0, 1))
        rf = ((rnd(Np, 2, nT, nC) if (nC > 1 or ri(0, 1)) else rnd(Np, 2, nT)) * 2 - 1) * 0.3
        gr = (rnd(Np, 3, nT) * 2 - 1) * 2
        loc = (rnd(N, nM, 3) * 2 - 1) * 8
        b1 = ((rnd(N, nM, 2, nC) * 2 - 1) if rf.ndim == 4 else (rnd(N, nM, 2) * 2 - 1)) if has_b1 else None
        df = ((rnd(N, nM) * 2 - 1) * 300) if ri(0, 1) else None
        relax = bool(ri(0, 2))
        T1, T2 = 0.3 + rnd(N, nM), 0.01 + 0.1 * rnd(N, nM)
        γ, dt = torch.tensor(4257.6, dtype=torch.float32), torch.tensor([4e-6], dtype=torch.float32)
        M0, w = rnd(N, nM, 3) * 2 - 1, rnd(N, nM, 3) * 2 - 1
        kw = dict(T1=T1, T2=T2) if relax else {}

        def run(kind):
            on = (lambda x: None if x is None else x.double()) if kind == 'oracle' else (lambda x: None if x is None else x.to(DEV))
            Mi, r, q = (on(x).clone().requires_grad_(True) for x in (M0, rf, gr))
            kk = {k_: on(v) for k_, v in kw.items()}
            if kind == 'oracle':
                Mo = O.blochsim(Mi, O.rfgr2beff(r, q, on(loc), Δf=on(df), b1Map=on(b1), γ=on(γ)), γ=on(γ), dt=on(dt), **kk)
            elif kind == 'two':
                Mo = sims.blochsim(Mi, beffective.rfgr2beff(r, q, on(loc), Δf=on(df), b1Map=on(b1), γ=on(γ)), γ=on(γ), dt=on(dt), **kk)
            else:
                Mo = fused.blochsim_rfgr(Mi, r, q, on(loc), Δf=on(df), b1Map=on(b1), γ_beff=on(γ), γ=on(γ), dt=on(dt), **kk)
            (Mo * on(w)).sum().backward()
            return Mi.grad, r.grad, q.grad
        ora = run('oracle')
        got = {kind: run(kind) for kind in ('two', 'fused')}
        tag = f'case {case}: N={N} nM={nM} nT={nT} nC={nC} Np={Np} b1={has_b1} df={df is not None} relax={relax}'
        if nT % 16 == 0 and nC <= 8:
            assert torch.equal(got['two'][0], got['fused'][0]), tag
        for kind, gs in got.items():
            for a, b, nm in zip(gs, ora, ('gMi', 'grf', 'ggr')):
                rel = float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
                worst[nm] = max(worst.get(nm, 0.0), rel)
                assert a.shape == b.shape and rel <= 3e-5, f'{tag} {kind} {nm} rel-L2 {rel:.2e}'
    print('fp32 gradient fuzz: worst relative L2', {k_: f'{v:.2e}' for k_, v in worst.items()})
