import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DT = {'f64': torch.float64, 'f32': torch.float32}
# Tolerances.  fp64: the reference's own (tests/test_sims.py:16).  fp32: BASELINE.json's
# north_star, "within 1e-5 rel fp32" -- tighter than the reference's own fp32 setting (1e-4,
# tests/test_sims.py:15); measured as relative L2 error over the tensor.
ATOL64 = 1e-9
REL32 = 1e-5


def golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def t(x, dtype=None, device='cpu'):
    y = torch.from_numpy(np.asarray(x))
    return y.to(device=device, dtype=dtype) if dtype is not None else y.to(device)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)


def max_abs(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max()) if a.numel() else 0.0


def assert_close(a, b, tag, what=''):
    r"""fp64: max-abs <= 1e-9 (reference's atol).  fp32: rel-L2 <= 1e-5 (north_star)."""
    if tag == 'f64':
        d = max_abs(a, b)
        assert d <= ATOL64, f'{what}: max abs diff {d:.3e} > {ATOL64:.0e}'
    else:
        d = rel_l2(a, b)
        assert d <= REL32, f'{what}: rel-L2 {d:.3e} > {REL32:.0e}'


def to_dev(d, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------
# Parity ledger: every distance a GPU test measures goes through record(); at the end of the
# session tests/conftest.py writes them to gpurun_out/parity_ledger.json (scratch; the judged copy
# is committed as profiles/rNN_parity.json).  Bounds in the tests cite these numbers.
# ---------------------------------------------------------------------------------------------
LEDGER = {}


def record(key: str, value, bound=None, note: str = None):
    r"""Note a measured distance (and the bound it is asserted against, if any); returns the value."""
    e = {'value': (float(value) if isinstance(value, (int, float)) or hasattr(value, '__float__') else value)}
    if bound is not None:
        e['bound'] = float(bound)
    if note:
        e['note'] = note
    LEDGER[key] = e
    return value


# Elementwise gates.  A relative L2 over six million numbers says nothing about the
# worst spin; the reference's own tolerance is elementwise -- ``pytest.approx(..., abs=atol)``, atol 1e-4 in fp32, on
# its 512-step case (/root/reference/tests/test_sims.py:15,101-105).
#
# What the worst spins are (profiles/r05_elementwise_scan.json, tools/elementwise_scan.py): spins whose field never
# changes direction -- the z = 0 plane of the synthetic cube, where the ramped z gradient contributes nothing.  Every
# step then makes the SAME rounding errors (b = γ2πdt·B in three components, x = b·b, S and C rounded to fp32), so they
# add up linearly instead of as a random walk: the rotation of each step is off by up to about one fp32 ulp of its angle,
# and after nT steps the phase is off by up to 2^-23 · Σ_t ϕ_t.  No fp32 step can do better on such a spin (the
# reference's sin/cos of the same ϕ every step do the same).  Measured, worst |Mo - exact| over that budget: HIP
# precise 0.33-0.40 on the 4096-spin subsets of configs[1] / [4] / [2], 0.67 / 0.64 over ALL spins of 64^3 x 1024 / x 2048,
# 0.27 over all 2 097 152 spins of the headline; the reference's own fp32 outputs 2.9-3.1.  Worst elements: HIP 3.9e-5 /
# 1.03e-4 / 6.3e-5 on the subsets, reference sims 4.7e-5 / 7.3e-5 / 1.9e-4, slowsims 5.2e-5 / 1.06e-4 / 1.8e-4; spins
# above 3e-5: HIP 2 / 5 / 7 of 4096, the reference 5-7 / 39-43 / 316-372.  So the gates are:
#   * per spin:   |Mo - exact| <= 2^-23 · Σ_t ϕ_t + 2e-6      (``angle_budget``; asserted for EVERY spin; grad_M0 likewise)
#   * the bulk:   at most 0.5 % of the spins above 3e-5, median <= 3e-6
#   * reference:  against the reference's own fp32 output, its own elementwise 1e-4 at nT <= 1024 (its own test is 512
#                 steps); beyond, the budget plus the reference's own worst distance from exact arithmetic.
# Gradients w.r.t. the pulse are sums over thousands of spins of any size: theirs is taken relative to the largest
# |element| of the yardstick (``scale=True``).
ATOL32_REFERENCE = 1e-4          # the reference's own fp32 setting: the outer gate
ELEM32_BULK = 3e-5               # at most BULK_FRACTION of the spins may be further than this from exact arithmetic
BULK_FRACTION = 0.005
ELEM32_MEDIAN = 3e-6
ELEM32_GRAD = 2e-5               # grad_rf / grad_gr, relative to the largest |element| of the exact gradient (measured: <= 4.2e-6;
                                 # the reference's own golden grad_gr: 1.7e-5)


def angle_budget(beff, γ2πdt, floor: float = 2e-6, chunk: int = 16384):
    r"""Per spin (flattened over ``(N, *Nd)``): ``2^-23 · Σ_t |γ2πdt · B_t| + floor`` -- the phase error a spin may have
    accumulated when every step's rotation is off by one fp32 ulp of its angle in the same direction.  ``γ2πdt``: 0-dim or
    uniform (its first element is used)."""
    b = torch.as_tensor(beff).detach()
    g = float(torch.as_tensor(γ2πdt).detach().double().reshape(-1)[0])
    rows = b.reshape(-1, b.shape[-2], 3)
    tot = torch.cat([(rows[i:i + chunk].double() * g).norm(dim=-1).sum(dim=-1).cpu() for i in range(0, rows.shape[0], chunk)])
    return tot * 2.0 ** -23 + floor


def elementwise(key: str, got, want, bound=None, *, scale: bool = False, comp_axis: int = -1, row_bound=None,
                bulk: bool = False):
    r"""Record ``<key>.max_abs`` (or ``.max_abs_over_max`` with ``scale``): the largest elementwise |got - want|, per
    component along ``comp_axis`` and overall, with the flat index of the worst row (spin / time point) -- and assert
    it against ``bound``.  ``row_bound`` (one number per row, e.g. :func:`angle_budget`): every row is asserted against
    its own bound and the worst ratio is recorded as ``<key>.max_abs_over_budget``.  ``bulk``: the count of rows above
    ``ELEM32_BULK`` and the median are recorded and asserted.  Returns the value."""
    a, b = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(want).detach().double().cpu()
    assert a.shape == b.shape, (key, a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    d = (a - b).abs().movedim(comp_axis, -1)
    denom = float(b.abs().max()) if scale else 1.0
    denom = denom if denom > 0 else 1.0
    flat = d.reshape(-1, d.shape[-1])
    rows = flat.max(dim=1).values
    per = (flat.max(dim=0).values / denom).tolist()
    row = int(rows.argmax())
    val = float(flat.max()) / denom
    k = key + ('.max_abs_over_max' if scale else '.max_abs')
    record(k, val, bound)
    view = lambda x: x.movedim(comp_axis, -1).reshape(-1, d.shape[-1])[row].tolist()  # noqa: E731
    LEDGER[k].update(per_component=[float(f'{x:.4e}') for x in per], worst_row=row,
                     worst_row_got=[float(x) for x in view(a)], worst_row_want=[float(x) for x in view(b)])
    if bound is not None:
        assert val <= bound, f'{k}: {val:.3e} > {bound:.1e} (worst row {row}, per component {per})'
    if row_bound is not None:
        rb = torch.as_tensor(row_bound).double().reshape(-1)
        assert rb.numel() == rows.numel(), (key, rb.numel(), rows.numel())
        ratio = rows / rb
        w = int(ratio.argmax())
        record(key + '.max_abs_over_budget', float(ratio.max()), 1.0,
               note=f'worst row {w}: |error| {float(rows[w]):.3e} against its budget 2^-23 x total rotation angle + floor '
                    f'= {float(rb[w]):.3e}')
        assert float(ratio.max()) <= 1.0, f'{key}: row {w} is {float(rows[w]):.3e} from the yardstick, budget {float(rb[w]):.3e}'
    if bulk:
        frac = float((rows > ELEM32_BULK).double().mean())
        record(key + '.fraction_of_rows_above_3e-5', frac, BULK_FRACTION, note=f'{int((rows > ELEM32_BULK).sum())} of {rows.numel()}')
        record(key + '.median_abs', float(rows.median()), ELEM32_MEDIAN)
        assert frac <= BULK_FRACTION and float(rows.median()) <= ELEM32_MEDIAN, (key, frac, float(rows.median()))
    return val
