"""K1h (reads Beff, writes the history) as a function of the RELATIVE placement of its two buffers:
one allocation holds [Beff | slack | history]; the history starts at Beff + S + delta (S = size of
Beff = 6 * 2^32 B at 128^3 x 1024, so delta is the distance modulo 2^32).  C ABI called directly.
    python tools/mode_probe3.py"""
import sys
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth, _host
n, nT = 128, 1024
dev = torch.device('cuda', 0)
lib = mrphy_amd.require_library()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
S = 12 * n ** 3 * nT
slack = 5 << 30
big = torch.empty(2 * S + slack, dtype=torch.uint8, device=dev)
base = big.data_ptr()
base += (-base) % (1 << 21)
beff = torch.empty(0)
P = beffective._PulseOnSpins(p['rf'], p['gr'], sp['loc'], sp['Δf'], None, sp['γ'])
st = _host.current_stream(dev)
assert lib.mrphy_rfgr2beff(0, *P.k0_args(), base, 1, n ** 3, nT, 1, st) == 0
g, E1, E2, E1_1 = sims.relax_constants(sp['T1'], sp['T2'], sp['γ'], p['dt'], 4, dev)
code, bg, e1, e2, e1m1 = sims._prep_constants(g, E1, E2, E1_1, 1, (n ** 3,), torch.float32, dev)
Mo = torch.empty_like(sp['M0'])
hb = int(lib.mrphy_blochsim_hist_bytes(code, 1, n ** 3, nT))
assert hb == S
print(f'Beff at 0x{base:x} (mod 2^32 = {base % (1 << 32)}), S = {S} = {S / 2**32} x 2^32', flush=True)
deltas = [0, 128, 1024, 4096, 65536, 1 << 20, 2 << 20, 6 << 20, 32 << 20, 128 << 20, 512 << 20, 1 << 30,
          (1 << 30) + (2 << 20), 2 << 30, (2 << 30) + 4096, 3 << 30, 4 << 30, (4 << 30) + 65536]
for d in deltas:
    hp = base + S + d
    ts = []
    for it in range(4):
        a, b = ev(), ev()
        a.record()
        rc = lib.mrphy_blochsim_fwd(code, sp['M0'].data_ptr(), base, *bg.args, *e1.args, *e2.args, e1m1.t.data_ptr(),
                                    Mo.data_ptr(), hp, 1, n ** 3, nT, st)
        b.record()
        assert rc == 0
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f'delta {d:11d} ({d / 2**20:9.3f} MiB)  K1h {min(ts[1:]):6.3f} ms = {2 * S / min(ts[1:]) / 1e9:5.2f} TB/s', flush=True)
