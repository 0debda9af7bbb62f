"""The bench's step (rfgr2beff, then blochsim on the block it has just written) for several builds of the library,
one child process each, twice:   python tools/ab_libs_step.py LIB_A.so LIB_B.so ..."""
import os, subprocess, sys
if len(sys.argv) >= 3 and sys.argv[1] != '--child':
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], capture_output=True, text=True)
            print(os.path.basename(lib), f'run {rep}:', r.stdout.strip() or r.stderr[-400:], flush=True)
    sys.exit(0)
lib = sys.argv[2]
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')]
import statistics, torch
import mrphy_amd
from mrphy_amd import _lib, beffective, sims, synth
_lib.library_path = lambda: os.path.abspath(lib)
dev = torch.device('cuda', 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
out = []
for label, n, nM, nT in (('cfg1', 64, 64 ** 3, 1024), ('shard', 128, 262144, 4096), ('128^3x1024', 128, 128 ** 3, 1024)):
    sp = synth.cube_spins(n, torch.arange(nM), dtype=torch.float32, device=dev)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    t0, t1 = [], []
    with torch.no_grad():
        blk = torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
        for rep in range(12):
            e = [ev() for _ in range(3)]
            e[0].record(); beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=blk)
            e[1].record(); Mo = sims.blochsim(sp['M0'], blk, **kw)
            e[2].record(); torch.cuda.synchronize()
            if rep >= 2:
                t0.append(e[0].elapsed_time(e[1])); t1.append(e[1].elapsed_time(e[2]))
    alg = 12 * nM * nT + 36 * nM
    out.append(f'{label}: K0 {statistics.median(t0):.3f} K1 {statistics.median(t1):.3f} ({alg / statistics.median(t1) / 8e9:.3f}) step {statistics.median(t0) + statistics.median(t1):.3f} |Mo| {float(Mo.double().norm()):.9e}')
    del blk, sp
print('  '.join(out))
