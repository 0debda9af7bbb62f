"""Which PAIRS of places make a fast two-part history?  (follow-up, for ABI 5, of round 5's window scans: profiles/r05_placement_windows_*.json)

One arena of ARENA_GB (a single allocation); the two parts of K1h's history (half the history each) are put at every
pair of offsets (i, j) of a grid with STEP_GB spacing, and K1h is timed (mrphy_blochsim_fwd_parts, blocked layout).
Also, per grid position: the whole history as ONE block starting there (the ABI <= 4 way).
If "fast" means "the two parts lie in different regions of the driver's memory", the matrix is block-structured.

    python tools/placement_pairs.py OUT.json [cube nT arena_GB step_GB]
"""
import ctypes
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import _lib  # noqa: E402

dev = torch.device('cuda', 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nT = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
arena_gb = float(sys.argv[4]) if len(sys.argv) > 4 else 96
step_gb = float(sys.argv[5]) if len(sys.argv) > 5 else 3
lib = mrphy_amd.require_library()
code, N, nM = _lib.F32P, 1, n ** 3
numel = nM * nT * 3
stream = torch.cuda.current_stream(dev).cuda_stream
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
whole = lib.mrphy_blochsim_hist_bytes(code, N, nM, nT) // 4
half = lib.mrphy_blochsim_hist_part_bytes(code, N, nM, nT, 2) // 4

field = torch.empty(numel, dtype=torch.float32, device=dev)
field.uniform_(-2.0, 2.0)
arena = torch.empty(int(arena_gb * (1 << 30)) // 4, dtype=torch.float32, device=dev)
Mi = torch.zeros((N, nM, 3), device=dev)
Mi[..., 2] = 1
Mo = torch.empty_like(Mi)
g = torch.tensor(2 * 3.141592653589793 * 4257.6 * 4e-6, device=dev)
E1 = torch.tensor(0.999996, device=dev)
E2 = torch.tensor(0.99994, device=dev)
E1m1 = E1 - 1
step = int(step_gb * (1 << 30)) // 4
base = arena.data_ptr()


def k1h(ptrs):
    tab = (ctypes.c_void_p * len(ptrs))(*ptrs)
    rc = lib.mrphy_blochsim_fwd_parts(code, Mi.data_ptr(), field.data_ptr(), g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                      E2.data_ptr(), 0, 0, E1m1.data_ptr(), Mo.data_ptr(), tab, len(ptrs), 0,
                                      N, nM, nT, stream)
    assert rc == 0, rc


def timed(fn, reps=2):
    fn()
    best = 1e9
    for _ in range(reps):
        a, b = ev(), ev()
        a.record(); fn(); b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(best, 4)


pos = []
off = 0
while off + half <= arena.numel():
    pos.append(off)
    off += step
one = []
for o in pos:
    one.append(timed(lambda: k1h([base + 4 * o])) if o + whole <= arena.numel() else None)
print('one block at each offset:', one, flush=True)
P = len(pos)
mat = [[None] * P for _ in range(P)]
for i in range(P):
    for j in range(P):
        if abs(pos[i] - pos[j]) >= half:
            mat[i][j] = timed(lambda: k1h([base + 4 * pos[i], base + 4 * pos[j]]))
    print(i, mat[i], flush=True)
json.dump({'cube': n, 'nT': nT, 'arena_GB': arena_gb, 'step_GB': step_gb, 'arena_ptr': hex(base),
           'device': torch.cuda.get_device_name(0), 'part_bytes': half * 4,
           'offsets_GiB': [o * 4 / (1 << 30) for o in pos], 'one_block_ms': one, 'pair_ms': mat},
          open(sys.argv[1], 'w'), indent=1)
