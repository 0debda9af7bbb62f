r"""Round-4 additions to the GPU suite (``-m gpu``, through the C ABI):

* ``rfgr2beff(..., out=)`` and the placement-aware ``workspace.BeffArena``: same bits as a fresh tensor, same
  gradients, misuse raises;
* K1's two tile orders (XCD-contiguous below 48 GB of ``Beff``, plain above) and its two schedules (pinned
  5-/6-step batches, unpinned 3-/4-step batches) are the same arithmetic: rows of a run on an XCD-ordered grid
  equal a run on those rows alone, bit for bit -- on grids whose tile count is not a multiple of 8 as well.
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
    assert len(calls) == 3 * 3                                 # one untimed + two timed launches per block
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
    r"""Below 48 GB of Beff the no-history K1 walks the spin tiles in XCD-contiguous order (block b -> tile
    (b % 8) * per_xcd + b / 8, grid padded to a multiple of 8): every row must still be integrated exactly once
    and exactly as on its own -- tile counts that are multiples of 8, not multiples of 8 (blocks past the last
    tile exit), and fewer than 8."""
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
