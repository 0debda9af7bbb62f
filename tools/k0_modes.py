"""Which of K0's two write rates (14.2 vs 16.7 ms at 128^3 x 4096) does a process get, and why?

Tests the placement hypothesis directly:
  A. several FRESH device allocations of the 103-GB output in one process (empty_cache between,
     a spacer allocation of varying size in front so that the physical placement moves): does the
     rate change between allocations of one process?
  B. one allocation with slack, output written at byte offsets 0 ... 1 GiB: does the rate depend on
     the pointer modulo the channel / XCD interleave?
  C. the same footprint written by torch's fill_ (a plain streaming write) and read back by K1
     next to every K0 timing: is it K0 or the box?

    python tools/k0_modes.py [cube] [nT]      (prints one line per measurement)
"""
import sys
import time

import torch

sys.path.insert(0, '.')
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, synth, _host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda', 0)
lib = mrphy_amd.require_library()
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
P = beffective._PulseOnSpins(p['rf'], p['gr'], sp['loc'], sp['Δf'], None, sp['γ'])
nbytes = P.N * P.nM * nT * 12


def ev():
    return torch.cuda.Event(enable_timing=True)


def k0_into(ptr, reps=5):
    ts = []
    for _ in range(reps + 1):
        a, b = ev(), ev()
        a.record()
        rc = lib.mrphy_rfgr2beff(0, *P.k0_args(), ptr, P.N, P.nM, nT, P.nC, _host.current_stream(dev))
        b.record()
        assert rc == 0
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:]), sum(ts[1:]) / reps


def fill_time(t, reps=3):
    ts = []
    for _ in range(reps + 1):
        a, b = ev(), ev()
        a.record()
        t.fill_(1.0)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


print(f'# workload {n}^3 x {nT}: Beff = {nbytes / 1e9:.2f} GB; spacing of the 8 XCD streams = '
      f'{nbytes // 8} B = {nbytes / 8 / 2**32:.4f} x 2^32', flush=True)

print('# A. fresh allocations (spacer GiB | ptr | ptr mod 2MiB, 1GiB | K0 min/avg ms | fill_ ms)')
import os  # noqa: E402
CHILD = bool(os.environ.get('K0_CHILD'))
for trial, spacer_gib in enumerate((0,) if CHILD else (0, 0, 1, 3, 7, 20)):
    torch.cuda.empty_cache()
    spacer = torch.empty(spacer_gib << 30, dtype=torch.uint8, device=dev) if spacer_gib else None
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    ptr = buf.data_ptr()
    mn, av = k0_into(ptr)
    f = fill_time(buf)
    print(f'A {trial}: spacer {spacer_gib:3d} GiB  ptr 0x{ptr:x}  mod2M {ptr % (2 << 20):8d}  '
          f'mod1G {ptr % (1 << 30):10d}  K0 {mn:7.3f} / {av:7.3f} ms = {nbytes / mn / 1e9:6.3f} TB/s  '
          f'fill_ {f:7.3f} ms = {nbytes / f / 1e9:6.3f} TB/s', flush=True)
    del buf, spacer

if CHILD:
    sys.exit(0)
print('# B. one allocation, output at byte offsets')
torch.cuda.empty_cache()
slack = 2 << 30
big = torch.empty(nbytes + slack, dtype=torch.uint8, device=dev)
base = big.data_ptr()
for off in (0, 128, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 64 << 20, 256 << 20, 1 << 30, (1 << 30) + (1 << 20)):
    mn, av = k0_into(base + off)
    print(f'B off {off:11d}  ptr mod 2^32 = {(base + off) % (1 << 32):11d}  K0 {mn:7.3f} / {av:7.3f} ms '
          f'= {nbytes / mn / 1e9:6.3f} TB/s', flush=True)
del big
torch.cuda.empty_cache()

print('# C. variants of the block order at the default placement')
import subprocess  # noqa: E402
if True:
    for v in ('0', '2021', '2020', '2041', '2081', '21', '1021'):
        env = dict(os.environ, K0_CHILD='1', MRPHY_K0_VARIANT=v)
        out = subprocess.run([sys.executable, __file__, str(n), str(nT)], env=env, capture_output=True,
                             text=True).stdout
        line = [ln for ln in out.splitlines() if ln.startswith('A 0')]
        print(f'C variant {v:>5}: {line[0] if line else out[-300:]}', flush=True)
