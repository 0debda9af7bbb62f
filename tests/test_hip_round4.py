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
