"""K1h / K3 with the history in ONE block or in separately allocated PARTS (ABI 5), in one fresh process.

VERDICT r5 item 3: the history is an internal buffer, nothing forces it to be one allocation.  The C entry points
`mrphy_blochsim_fwd_parts / _bwd_parts` are called directly on blocks allocated in a known order, so that what is
timed is the kernels on a given placement, not the caching allocator's choices:

    group k = 0 .. G-1, allocated in this order:   S_k (whole history)   A_k (part)   X_k (grad block)   B_k (part)
                                                    C_k (part)  D_k (part)      [C_k, D_k are neighbours]

    K1h:  one part  S_k                        (what ABI <= 4 did)
          two parts (A_k, B_k)  blocked / interleaved      -- parts separated by another allocation
          two parts (C_k, D_k)  blocked                    -- neighbouring allocations
          two parts (A_k, B_(k+G/2)) blocked               -- far apart
          four parts (A_k, B_k, C_k, D_k) at half length each, blocked
    K3 :  history in S_k or in (A_k, B_k), grad_Beff written to X_k or to S_((k+1) % G)

    python tools/hist_parts_ab.py OUT.json [--n 64] [--nT 2048] [--groups 5] [--f64]
"""
import argparse
import ctypes
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('out')
ap.add_argument('--n', type=int, default=64)
ap.add_argument('--nT', type=int, default=2048)
ap.add_argument('--groups', type=int, default=5)
ap.add_argument('--reps', type=int, default=4)
ap.add_argument('--f64', action='store_true')
a = ap.parse_args()

dev = torch.device('cuda', 0)
lib = mrphy_amd.require_library()
dtype = torch.float64 if a.f64 else torch.float32
code = _lib.F64 if a.f64 else _lib.F32P
es = 8 if a.f64 else 4
N, nM, nT = 1, a.n ** 3, a.nT
numel = nM * nT * 3
stream = torch.cuda.current_stream(dev).cuda_stream
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

whole = lib.mrphy_blochsim_hist_bytes(code, N, nM, nT) // es
half = lib.mrphy_blochsim_hist_part_bytes(code, N, nM, nT, 2) // es
quarter = lib.mrphy_blochsim_hist_part_bytes(code, N, nM, nT, 4) // es
new = lambda k: torch.empty(k, dtype=dtype, device=dev)  # noqa: E731

field = new(numel)
field.uniform_(-2.0, 2.0)
Mi = torch.zeros((N, nM, 3), dtype=dtype, device=dev)
Mi[..., 2] = 1
Mo, gMi = torch.empty_like(Mi), torch.empty_like(Mi)
gMo = torch.ones_like(Mi)
cdt = dtype
g = torch.tensor(2 * 3.141592653589793 * 4257.6 * 4e-6, dtype=cdt, device=dev)
E1 = torch.tensor(0.999996, dtype=cdt, device=dev)
E2 = torch.tensor(0.99994, dtype=cdt, device=dev)
E1m1 = E1 - 1

S, A, X, B, C, D = [], [], [], [], [], []
for k in range(a.groups):
    S.append(new(whole)); A.append(new(half)); X.append(new(numel)); B.append(new(half))
    C.append(new(half)); D.append(new(half))


def table(parts):
    return (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts]), len(parts)


def k1h(parts, layout=0):
    tab, n = table(parts)
    rc = lib.mrphy_blochsim_fwd_parts(code, Mi.data_ptr(), field.data_ptr(), g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                      E2.data_ptr(), 0, 0, E1m1.data_ptr(), Mo.data_ptr(), tab, n, layout,
                                      N, nM, nT, stream)
    assert rc == 0, rc


def k3(parts, gB, layout=0):
    tab, n = table(parts)
    rc = lib.mrphy_blochsim_bwd_parts(code, tab, n, layout, field.data_ptr(), g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                      E2.data_ptr(), 0, 0, gMo.data_ptr(), gMi.data_ptr(), gB.data_ptr(), None,
                                      N, nM, nT, stream)
    assert rc == 0, rc


def timed(fn):
    fn()
    ts = []
    for _ in range(a.reps):
        e0, e1 = ev(), ev()
        e0.record(); fn(); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return round(ts[0], 4), round(ts[len(ts) // 2], 4)


rows = []
k1h_bytes = 24 * nM * nT * (es // 4)
k3_bytes = 36 * nM * nT * (es // 4)


def rec(kernel, kind, k, t, **kw):
    b = k1h_bytes if kernel == 'K1h' else k3_bytes
    r = dict(kernel=kernel, kind=kind, group=k, ms_min=t[0], ms_med=t[1], frac_hbm=round(b / (t[0] * 1e-3) / 8e12, 3), **kw)
    print(json.dumps(r), flush=True)
    rows.append(r)


G = a.groups
# reference results for the bit-identity check
k1h([S[0]]); k3([S[0]], X[0])
torch.cuda.synchronize()
Mo_ref, gMi_ref, gB_ref = Mo.clone(), gMi.clone(), X[0].clone()
same = True
for k in range(G):
    rec('K1h', 'one block', k, timed(lambda: k1h([S[k]])))
    rec('K1h', 'two parts, another allocation between, blocked', k, timed(lambda: k1h([A[k], B[k]], 0)))
    rec('K1h', 'two parts, another allocation between, interleaved', k, timed(lambda: k1h([A[k], B[k]], 1)))
    rec('K1h', 'two parts, neighbours, blocked', k, timed(lambda: k1h([C[k], D[k]], 0)))
    rec('K1h', 'two parts, far apart, blocked', k, timed(lambda: k1h([A[k], B[(k + G // 2) % G]], 0)))
    q = [A[k][:quarter], B[k][:quarter], C[k][:quarter], D[k][:quarter]]
    rec('K1h', 'four parts, blocked', k, timed(lambda: k1h(q, 0)))
    rec('K1h', 'four parts, interleaved', k, timed(lambda: k1h(q, 1)))
for k in range(G):
    k1h([S[k]])
    rec('K3', 'history one block, grad_Beff -> X', k, timed(lambda: k3([S[k]], X[k])))
    k1h([A[k], B[k]], 0)
    rec('K3', 'history two parts, grad_Beff -> X', k, timed(lambda: k3([A[k], B[k]], X[k], 0)))
    torch.cuda.synchronize()
    same = same and torch.equal(X[k][:numel], gB_ref[:numel]) and torch.equal(gMi, gMi_ref) and torch.equal(Mo, Mo_ref)
    k1h([A[k], B[k]], 1)
    rec('K3', 'history two parts interleaved, grad_Beff -> X', k, timed(lambda: k3([A[k], B[k]], X[k], 1)))
    torch.cuda.synchronize()
    same = same and torch.equal(X[k][:numel], gB_ref[:numel])
    k1h([A[k], B[k]], 0)
    rec('K3', 'history two parts, grad_Beff -> S (a whole-history block)', k,
        timed(lambda: k3([A[k], B[k]], S[(k + 1) % G], 0)))
print('bit-identical to the one-part route:', same, flush=True)
json.dump({'cube': a.n, 'nT': nT, 'dtype': str(dtype), 'groups': G, 'bit_identical': bool(same),
           'device': torch.cuda.get_device_name(0),
           'ptr': {nm: [hex(t.data_ptr()) for t in L] for nm, L in (('S', S), ('A', A), ('X', X), ('B', B), ('C', C), ('D', D))},
           'rows': rows}, open(a.out, 'w'), indent=1)
