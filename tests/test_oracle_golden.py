r"""The CPU oracle against the committed golden vectors (outputs of the reference itself, made
by tests/golden/make_golden.py) and against the known answers hard-coded in the reference's own
tests.  Runs without a GPU."""
import os

import numpy as np
import pytest
import torch

import bloch_oracle as O
import cases
from util import DT, golden, t, assert_close, max_abs, rel_l2

# Known answers the reference's tests hold (tests/test_slowsims.py:77-80 == test_mobjs.py:112-115;
# no-relaxation: tests/test_mobjs.py:117-120).  Data, fp64, atol 1e-9.
MO0_RELAX = np.array([[[0.559535641648385, 0.663342640621335, 0.416341441715101],
                       [0.391994737048090, 0.210182892388552, -0.860954821972489],
                       [-0.677062008711222, 0.673391604920576, -0.143262993311057]]])
MO0_NORELAX = np.array([[[0.584337330324116, 0.686096989146395, 0.433382978292808],
                         [0.404188676945936, 0.217027890590635, -0.888555236400348],
                         [-0.703691265981316, 0.694384487290747, -0.150495136106067]]])


def _beff(c):
    return O.rfgr2beff(c['rf'], c['gr'], c['loc'], Δf=c['Δf'], b1Map=c['b1Map'], γ=c['γ'])


def test_known_answers_fp64():
    c = cases.ref_case(3, torch.float64)
    beff = _beff(c)
    kw = dict(γ=c['γ'], dt=c['dt'])
    for fn in (O.blochsim_slow, O.blochsim):
        assert max_abs(fn(c['M0'], beff, T1=c['T1'], T2=c['T2'], **kw), MO0_RELAX) < 1e-9
        assert max_abs(fn(c['M0'], beff, **kw), MO0_NORELAX) < 1e-9
    # 512 x blochsim_1step == blochsim (test_slowsims.py:65-69)
    E1, E2 = torch.exp(-c['dt'] / c['T1']), torch.exp(-c['dt'] / c['T2'])
    g = 2 * np.pi * c['γ'] * c['dt']
    M = c['M0'].clone()
    for i in range(beff.shape[-2]):
        M, _ = O.blochsim_1step(M, None, beff[..., i, :], E1, E1 - 1, E2, g)
    assert max_abs(M, MO0_RELAX) < 1e-9
    # golden file agrees with the hard-coded numbers too
    G = golden('ref3_f64')
    assert max_abs(G['Mo_slow'], MO0_RELAX) < 1e-9 and max_abs(G['Mo_sims_norelax'], MO0_NORELAX) < 1e-9


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_ref3_golden(tag):
    G, c = golden(f'ref3_{tag}'), cases.ref_case(3, DT[tag])
    beff = _beff(c)
    assert max_abs(beff, G['beff']) == 0.0                      # same op sequence: bit-exact
    kw = dict(γ=c['γ'], dt=c['dt'])
    assert max_abs(O.blochsim_slow(c['M0'], beff, T1=c['T1'], T2=c['T2'], **kw), G['Mo_slow']) == 0.0
    assert max_abs(O.blochsim(c['M0'], beff, T1=c['T1'], T2=c['T2'], **kw), G['Mo_sims']) == 0.0
    assert max_abs(O.blochsim(c['M0'], beff, **kw), G['Mo_sims_norelax']) == 0.0
    rf, gr = c['rf'].clone().requires_grad_(True), c['gr'].clone().requires_grad_(True)
    b = O.rfgr2beff(rf, gr, c['loc'], Δf=c['Δf'], b1Map=c['b1Map'], γ=c['γ'])
    O.blochsim(c['M0'], b, T1=c['T1'], T2=c['T2'], **kw).sum().backward()
    assert_close(rf.grad, G['grad_rf'], tag, 'grad_rf')
    assert_close(gr.grad, G['grad_gr'], tag, 'grad_gr')


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_ref512_golden(tag):
    G, c = golden(f'ref512_{tag}'), cases.ref_case(512, DT[tag], seed=1234)
    assert max_abs(c['M0'], G['M0']) == 0.0, 'seeded M0 differs: torch CPU RNG changed'
    beff = _beff(c)
    rows = G['rows'].tolist()
    assert max_abs(beff[:, rows], G['beff_rows']) == 0.0
    assert float(G['beff_nodim_maxdiff']) <= (1e-9 if tag == 'f64' else 1e-4)
    for relax in (True, False):
        rk = dict(T1=c['T1'], T2=c['T2']) if relax else {}
        sfx = '' if relax else '_norelax'
        for name, fn in (('slow', O.blochsim_slow), ('sims', O.blochsim)):
            M0 = c['M0'].clone().requires_grad_(True)
            B = beff.clone().requires_grad_(True)
            Mo = fn(M0, B, **rk, γ=c['γ'], dt=c['dt'])
            Mo.sum().backward()
            assert max_abs(Mo, G[f'Mo_{name}{sfx}']) == 0.0
            assert_close(M0.grad, G[f'gM0_{name}{sfx}'], tag, f'gM0 {name}{sfx}')
            assert_close(B.grad[:, rows], G[f'gB_rows_{name}{sfx}'], tag, f'gB {name}{sfx}')
            assert_close(B.grad.sum(1), G[f'gB_sum_{name}{sfx}'], tag, f'gB sum {name}{sfx}')
        # the reference's own test: sims and slowsims gradients agree (test_sims.py:104-105)
        tol = 1e-9 if tag == 'f64' else 1e-4
        assert max_abs(G[f'gM0_sims{sfx}'], G[f'gM0_slow{sfx}']) < tol
        assert max_abs(G[f'gB_rows_sims{sfx}'], G[f'gB_rows_slow{sfx}']) < tol


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_rfgr_variants_golden(tag):
    G = golden(f'rfgr_{tag}')
    for name, kw in cases.rfgr_variants(DT[tag]).items():
        kw = dict(kw)
        rf, gr, loc = kw.pop('rf'), kw.pop('gr'), kw.pop('loc')
        # requires_grad as in the generator: torch.matmul folds a broadcast batch differently
        # with and without autograd (1 ulp), for the reference and the oracle alike
        rf, gr = rf.clone().requires_grad_(True), gr.clone().requires_grad_(True)
        beff = O.rfgr2beff(rf, gr, loc, **kw)
        assert max_abs(beff.detach(), G[f'{name}.beff']) == 0.0, name
        w = torch.cos(torch.arange(beff.numel(), dtype=torch.float64) * 0.37).reshape(beff.shape)
        (beff * w.to(DT[tag])).sum().backward()
        assert max_abs(rf.grad, G[f'{name}.grad_rf']) == 0.0 and \
            max_abs(gr.grad, G[f'{name}.grad_gr']) == 0.0, name


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_bcast_golden(tag):
    G = golden(f'bcast_{tag}')
    M0, Beff, variants = cases.bcast_variants(DT[tag])
    for name, kw in variants.items():
        Mi, B = M0.clone().requires_grad_(True), Beff.clone().requires_grad_(True)
        Mo = O.blochsim(Mi, B, **kw)
        w = torch.sin(torch.arange(Mo.numel(), dtype=torch.float64) * 0.61 + 1).reshape(Mo.shape)
        (Mo * w.to(DT[tag])).sum().backward()
        assert max_abs(Mo, G[f'{name}.Mo']) == 0.0, name
        assert_close(B.grad, G[f'{name}.gB'], tag, f'{name}.gB')
        if f'{name}.gMi' in G:
            assert_close(Mi.grad, G[f'{name}.gMi'], tag, f'{name}.gMi')
        # explicit adjoint == autograd of the out-of-place form, also where the reference's
        # grad_Mi is broken (per-spin γ, per-batch dt)
        if kw['dt'].numel() == 1:      # slowsims needs a batch-uniform dt (slowsims.py:90)
            Mi2, B2 = M0.clone().requires_grad_(True), Beff.clone().requires_grad_(True)
            k2 = {k: (v.to(DT[tag]) if v is not None else None) for k, v in kw.items()}
            (O.blochsim_slow(Mi2, B2, **k2) * w.to(DT[tag])).sum().backward()
            # ... except at exactly-zero field, where autograd differentiates norm() at 0 to a
            # zero subgradient while the explicit adjoint (reference sims.py and this oracle)
            # returns the analytic limit -γ2πdt·(m×h̃): compare the non-zero-field samples.
            nz = (Beff != 0).any(dim=-1, keepdim=True).expand_as(Beff)
            live = nz.all(dim=-1).all(dim=-1)             # spins that are never at zero field
            assert_close(Mi.grad[live], Mi2.grad[live], tag, f'{name}: explicit vs autograd gMi')
            assert_close(B.grad[live], B2.grad[live], tag, f'{name}: explicit vs autograd gB')


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_onestep_uphi_golden(tag):
    G, U = golden(f'onestep_{tag}'), golden(f'uphi_{tag}')
    c = cases.onestep_case(DT[tag])
    Mn, Mold = O.blochsim_1step(c['M'].clone(), None, c['b'], c['E1'], c['E1_1'], c['E2'], c['γ2πdt'])
    assert max_abs(Mn, G['M_new']) == 0.0
    Min = c['M'].clone()
    Mz, _ = O.blochsim_1step(Min, None, torch.zeros_like(c['b']), c['E1'], c['E1_1'], c['E2'], c['γ2πdt'])
    assert max_abs(Mz, G['M_new_zero_b']) == 0.0
    assert Mz is Min                       # all-zero field: the input itself is relaxed in place
    u, p = O.beff2uphi(c['b'], c['γ2πdt'])
    assert max_abs(u, U['U']) <= (1e-15 if tag == 'f64' else 1e-7) and max_abs(p, U['Phi']) == 0.0
    V34 = torch.stack([c['M'], c['M'].flip(-1), c['M'] * 2, -c['M']], dim=-1)
    assert max_abs(O.uphirot(t(U['U']), t(U['Phi']), c['M']), U['rot3']) == 0.0
    assert max_abs(O.uphirot(t(U['U']), t(U['Phi']), V34), U['rot34']) == 0.0


def test_big_subset_golden_cfg1():
    r"""64^3 x 1024 config, 4096-spin subset: oracle == reference rows, bit for bit."""
    G = golden('big_cfg1_f32')
    idx, sp, pulse = cases.big_subset(1, torch.float32, 4096)
    assert np.array_equal(idx.numpy(), G['idx']) and max_abs(sp['M0'], G['M0']) == 0.0
    n = 512                                   # keep the CPU suite short
    beff = O.rfgr2beff(pulse['rf'], pulse['gr'], sp['loc'][:, :n], Δf=sp['Δf'][:, :n], γ=sp['γ'])
    Mo = O.blochsim(sp['M0'][:, :n], beff, T1=sp['T1'][:, :n], T2=sp['T2'][:, :n], γ=sp['γ'],
                    dt=pulse['dt'])
    assert max_abs(Mo, G['Mo_sims'][:, :n]) == 0.0
    # the reference's own two fp32 implementations differ by this much on this workload:
    assert rel_l2(G['Mo_sims'], G['Mo_slow']) < 2e-5


def test_f64_arith_yardstick():
    r"""fp64 arithmetic with the fp32-rounded constants reproduces fp64 results when the inputs
    are fp64, and stays within fp32 round-off of the fp32 run."""
    c = cases.ref_case(3, torch.float64)
    beff = _beff(c)
    kw = dict(T1=c['T1'], T2=c['T2'], γ=c['γ'], dt=c['dt'])
    assert max_abs(O.blochsim_f64_arith(c['M0'], beff, **kw), MO0_RELAX) < 1e-9
    c32 = cases.ref_case(3, torch.float32)
    b32 = _beff(c32)
    k32 = dict(T1=c32['T1'], T2=c32['T2'], γ=c32['γ'], dt=c32['dt'])
    assert rel_l2(O.blochsim(c32['M0'], b32, **k32), O.blochsim_f64_arith(c32['M0'], b32, **k32)) < 1e-5


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_freeprec_golden(tag):
    r"""freeprec: the reference's known answer (tests/test_slowsims.py:100-121, test_mobjs.py:152)
    and the golden variants, forward and grad_Mi, explicit and autograd forms."""
    G = golden(f'freeprec_{tag}')
    known = np.array([[[0., -0.5, 0.5], [-0.5, 0, 0.5], [0., 0., 1.]]])
    for name, kw in cases.freeprec_variants(DT[tag]).items():
        kw = dict(kw)
        M, dur = kw.pop('M'), kw.pop('dur')
        for impl, fn in (('sims', O.freeprec), ('slow', O.freeprec_slow)):
            Mi = M.clone().requires_grad_(True)
            Mo = fn(Mi, dur, **kw)
            w = torch.cos(torch.arange(Mo.numel(), dtype=torch.float64) * 0.53).reshape(Mo.shape)
            (Mo * w.to(DT[tag])).sum().backward()
            assert max_abs(Mo, G[f'{name}.Mo_{impl}']) == 0.0, (name, impl)
            assert_close(Mi.grad, G[f'{name}.gMi_{impl}'], tag, f'{name}.gMi_{impl}')
            if name == 'known':
                assert max_abs(Mo, known) < (1e-9 if tag == 'f64' else 1e-6)


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_masks_golden(tag):
    r"""the oracle's mask gather/scatter and cube locations reproduce the reference's
    SpinArray.extract/embed and SpinCube._update_loc_ outputs bit for bit (NaN pattern included)."""
    G, c = golden(f'masks_{tag}'), cases.mask_case(DT[tag])
    assert np.array_equal(G['mask'], c['mask'].numpy())
    for name, v in c['spatial'].items():
        assert np.array_equal(O.mask_extract(v, c['mask']).numpy(), G[f'extract.{name}'])
    for name, v_ in c['compact'].items():
        assert np.array_equal(O.mask_embed(v_, c['mask']).numpy(), G[f'embed.{name}'], equal_nan=True)
    got = O.mask_embed(c['compact']['M'], c['mask'], out=c['spatial']['M'].clone())
    assert np.array_equal(got.numpy(), G['embed_out.M'])
    assert np.array_equal(O.cube_loc(c['mask'], c['fov'], c['ofst']).numpy(), G['loc_'])
    # and the mobjs fixture of the reference's own test (test_mobjs.py:98-131): its loc_
    M = golden(f'mobjs_{tag}')
    fov, ofst = torch.tensor([[3., 3., 3.]], dtype=DT[tag]), torch.tensor([[0., 0., 1.]], dtype=DT[tag])
    assert np.array_equal(O.cube_loc(torch.from_numpy(M['mask']), fov, ofst).numpy(), M['loc_'])


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_ab_golden(tag):
    r"""oracle beff2ab / blochsim_ab vs the reference's outputs on its own 3-spin
    case (known answer test_slowsims.py:77-80 included) and on the 512-spin line."""
    G, c = golden(f'ab3_{tag}'), cases.ref_case(3, DT[tag])
    beff, E1, E2 = t(G['beff']), t(G['E1']), t(G['E2'])
    A, B = O.beff2ab(beff, E1=E1, E2=E2, γ=c['γ'], dt=c['dt'])
    assert np.array_equal(A.numpy(), G['A']) and np.array_equal(B.numpy(), G['B'])
    Mo = O.blochsim_ab(c['M0'], A, B)
    assert np.array_equal(Mo.numpy(), G['Mo'])
    if tag == 'f64':
        assert max_abs(Mo, MO0_RELAX) <= 1e-9
    A0, B0 = O.beff2ab(beff, γ=c['γ'], dt=c['dt'])
    assert np.array_equal(A0.numpy(), G['A_E0']) and np.array_equal(B0.numpy(), G['B_E0'])
    # affine-map property: A M + B == stepping M through the pulse
    want = O.blochsim_slow(c['M0'].clone(), beff, T1=c['T1'], T2=c['T2'], γ=c['γ'], dt=c['dt'])
    assert_close(Mo, want, tag, 'A M + B vs blochsim')
    G5, c5 = golden(f'ab512_{tag}'), cases.ref_case(512, DT[tag], seed=1234)
    b5 = O.rfgr2beff(c5['rf'], c5['gr'], c5['loc'], Δf=c5['Δf'], b1Map=c5['b1Map'], γ=c5['γ'])
    A5, B5 = O.beff2ab(b5, E1=t(G5['E1']), E2=t(G5['E2']), γ=c5['γ'], dt=c5['dt'])
    assert_close(A5, G5['A'], tag, 'A 512')
    assert_close(B5, G5['B'], tag, 'B 512')


def test_c_restatement():
    r"""oracle/bloch_c.c (plain C, fp64, the reference's axis/angle form) against the reference's
    known answers and golden outputs, and against the torch restatement on the multi-coil /
    batched variants: two independent restatements pinned by the same vectors."""
    import bloch_c as C
    G, c = golden('ref3_f64'), cases.ref_case(3, torch.float64)
    b = C.rfgr2beff(c['rf'], c['gr'], c['loc'], Δf=c['Δf'], b1Map=c['b1Map'], γ=c['γ'])
    assert max_abs(b, G['beff']) <= 1e-12
    kw = dict(γ=c['γ'], dt=c['dt'])
    Mo = C.blochsim(c['M0'], b, T1=c['T1'], T2=c['T2'], **kw)
    assert max_abs(Mo, MO0_RELAX) <= 1e-9 and max_abs(Mo, G['Mo_slow']) <= 1e-12
    assert max_abs(C.blochsim(c['M0'], b, **kw), MO0_NORELAX) <= 1e-9
    Mf = C.blochsim_rfgr(c['M0'], c['rf'], c['gr'], c['loc'], Δf=c['Δf'], b1Map=c['b1Map'],
                         γ_beff=c['γ'], T1=c['T1'], T2=c['T2'], **kw)
    assert max_abs(Mf, Mo) == 0.0
    G5, c5 = golden('ref512_f64'), cases.ref_case(512, torch.float64, seed=1234)
    b5 = C.rfgr2beff(c5['rf'], c5['gr'], c5['loc'], Δf=c5['Δf'], b1Map=c5['b1Map'], γ=c5['γ'])
    assert max_abs(C.blochsim(c5['M0'], b5, T1=c5['T1'], T2=c5['T2'], γ=c5['γ'], dt=c5['dt']),
                   G5['Mo_sims']) <= 1e-9
    for name, v in cases.rfgr_variants(torch.float64).items():
        v = dict(v)
        rf, gr, loc = v.pop('rf'), v.pop('gr'), v.pop('loc')
        assert max_abs(C.rfgr2beff(rf, gr, loc, **v), O.rfgr2beff(rf, gr, loc, **v)) <= 1e-11, name
    M0, Beff, variants = cases.bcast_variants(torch.float64)
    for name, kw in variants.items():
        assert max_abs(C.blochsim(M0, Beff, **kw), O.blochsim(M0, Beff, **kw)) <= 1e-11, name


def test_c_restatement_adjoint():
    r"""The explicit adjoint of oracle/bloch_c.c (sims.py:135-269 in double) against the reference's
    golden gradients -- ``grad_M0``, rows and spin-sum of ``grad_beff`` from both reference
    implementations (what the reference itself tests: tests/test_sims.py:104-105,142-143), the
    gradient chain to ``rf``/``gr`` (tests/test_slowsims.py:86-96) -- and, on the multi-coil,
    batched and per-spin-constant variants, against the torch restatement's autograd."""
    import bloch_c as C
    G, c = golden('ref512_f64'), cases.ref_case(512, torch.float64, seed=1234)
    beff = _beff(c)
    rows = G['rows'].tolist()
    ones = torch.ones_like(c['M0'])
    for relax in (True, False):
        rk = dict(T1=c['T1'], T2=c['T2']) if relax else {}
        sfx = '' if relax else '_norelax'
        gMi, gB = C.blochsim_bwd(c['M0'], beff, ones, **rk, γ=c['γ'], dt=c['dt'])
        for ref in ('sims', 'slow'):
            assert max_abs(gMi, G[f'gM0_{ref}{sfx}']) <= 1e-9
            assert max_abs(gB[:, rows], G[f'gB_rows_{ref}{sfx}']) <= 1e-9
            assert max_abs(gB.sum(1), G[f'gB_sum_{ref}{sfx}']) <= 1e-9
    # the fused form: chain rule to rf / gr (reference goldens of the 3-spin case, fp64)
    G3, c3 = golden('ref3_f64'), cases.ref_case(3, torch.float64)
    kw3 = dict(Δf=c3['Δf'], b1Map=c3['b1Map'], γ_beff=c3['γ'], T1=c3['T1'], T2=c3['T2'], γ=c3['γ'],
               dt=c3['dt'])
    Mo, gMi, grf, ggr = C.blochsim_rfgr_grad(c3['M0'], c3['rf'], c3['gr'], c3['loc'], **kw3)
    assert max_abs(Mo, MO0_RELAX) <= 1e-9
    assert max_abs(grf.reshape(G3['grad_rf'].shape), G3['grad_rf']) <= 1e-9
    assert max_abs(ggr, G3['grad_gr']) <= 1e-9
    # variants (multi-coil b1Map, N = 2, no Δf ...) with a non-trivial cotangent, vs torch autograd
    for name, v in cases.rfgr_variants(torch.float64).items():
        v = dict(v)
        rf, gr, loc = v.pop('rf'), v.pop('gr'), v.pop('loc')
        N, nM = loc.shape[0], loc.shape[1]
        g = torch.Generator().manual_seed(3)
        M0 = torch.rand((N, nM, 3), generator=g, dtype=torch.float64)
        w = torch.rand((N, nM, 3), generator=g, dtype=torch.float64) - 0.5
        T1 = 0.5 + torch.rand((N, nM), generator=g, dtype=torch.float64)
        T2 = 0.03 + 0.1 * torch.rand((N, nM), generator=g, dtype=torch.float64)
        γ = v.get('γ', O.γH)
        rfo, gro = rf.clone().requires_grad_(True), gr.clone().requires_grad_(True)
        M0o = M0.clone().requires_grad_(True)
        Mo_o = O.blochsim(M0o, O.rfgr2beff(rfo, gro, loc, **v), T1=T1, T2=T2, γ=γ, dt=O.dt0)
        (Mo_o * w).sum().backward()
        Mo, gMi, grf, ggr = C.blochsim_rfgr_grad(M0, rf, gr, loc, w, Δf=v.get('Δf'),
                                                 b1Map=v.get('b1Map'), γ_beff=γ, T1=T1, T2=T2, γ=γ,
                                                 dt=O.dt0)
        assert max_abs(Mo, Mo_o) <= 1e-11 and max_abs(gMi, M0o.grad) <= 1e-10, name
        want_rf, want_gr = rfo.grad, gro.grad
        if rf.ndim == 4 and v.get('b1Map') is None:      # summed over coils before use
            want_rf = want_rf[..., 0]
        got_rf = grf if grf.shape[0] == want_rf.shape[0] else grf.sum(0, keepdim=True)
        got_gr = ggr if ggr.shape[0] == want_gr.shape[0] else ggr.sum(0, keepdim=True)
        scale = max(1.0, float(want_rf.abs().max()), float(want_gr.abs().max()))
        assert max_abs(got_rf, want_rf) <= 1e-9 * scale, name
        assert max_abs(got_gr, want_gr) <= 1e-9 * scale, name


@pytest.mark.parametrize('cfg', [1, 2])
def test_oracle_single_precision_field_is_the_reference_beff(cfg):
    r"""Round 4: the C restatement's single-precision field -- what its ``field_f32=True`` integrations
    and gradients use as "the same fp32 field" -- equals the reference's own fp32 ``rfgr2beff`` rows bit for bit
    (``big_beff_rows_f32.npz``: eight spins of each BASELINE config, written by ``make_golden.py`` from the
    imported reference).  The GPU suite asserts the same of K0."""
    import numpy as np
    import cases
    import bloch_c as C
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'big_beff_rows_f32.npz'),
                 allow_pickle=False) as z:
        idx, want = torch.from_numpy(z[f'cfg{cfg}.idx']), torch.from_numpy(z[f'cfg{cfg}.beff'])
    idx_all, sp, pulse = cases.big_subset(cfg, torch.float32, 4096)
    assert torch.equal(idx_all[:idx.numel()], idx)
    sl = slice(0, idx.numel())
    f = C.field_f32(pulse['rf'], pulse['gr'], sp['loc'][:, sl], Δf=sp['Δf'][:, sl], γ_beff=sp['γ'])
    assert torch.equal(f, want)
