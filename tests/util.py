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
# worst spin; the reference's own tolerance is elementwise -- ``pytest.approx(..., abs=atol)``, atol 1e-4 in fp32
# (/root/reference/tests/test_sims.py:15,101-105).  |Mo| <= 1, so for magnetisations the max abs error is also the
# error relative to full scale; gradients are sums over thousands of spins of any size, so theirs is taken relative to
# the largest |element| of the yardstick (``scale=True``).
ATOL32_REFERENCE = 1e-4          # the reference's own fp32 setting: the outer gate
# Tighter bounds proposed from the measured worst elements on MI355X (profiles/r05_parity.json; about 3 x measured):
ELEM32_MO = 3e-5                 # Mo vs exact arithmetic on the same fp32 field and constants, nT <= 4096
ELEM32_GRAD = 3e-5               # gradients, relative to the largest |element| of the exact gradient


def elementwise(key: str, got, want, bound=None, *, scale: bool = False, comp_axis: int = -1):
    r"""Record ``<key>.max_abs`` (or ``.max_abs_over_max`` with ``scale``): the largest elementwise |got - want|, per
    component along ``comp_axis`` and overall, with the flat index of the worst row (spin / time point) -- and assert
    it against ``bound``.  Returns the value."""
    a, b = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(want).detach().double().cpu()
    assert a.shape == b.shape, (key, a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    d = (a - b).abs().movedim(comp_axis, -1)
    denom = float(b.abs().max()) if scale else 1.0
    denom = denom if denom > 0 else 1.0
    flat = d.reshape(-1, d.shape[-1])
    per = (flat.max(dim=0).values / denom).tolist()
    row = int(flat.max(dim=1).values.argmax())
    val = float(flat.max()) / denom
    k = key + ('.max_abs_over_max' if scale else '.max_abs')
    record(k, val, bound)
    LEDGER[k].update(per_component=[float(f'{x:.4e}') for x in per], worst_row=row,
                     worst_row_got=[float(x) for x in a.movedim(comp_axis, -1).reshape(-1, d.shape[-1])[row].tolist()],
                     worst_row_want=[float(x) for x in b.movedim(comp_axis, -1).reshape(-1, d.shape[-1])[row].tolist()])
    if bound is not None:
        assert val <= bound, f'{k}: {val:.3e} > {bound:.1e} (worst row {row}, per component {per})'
    return val
