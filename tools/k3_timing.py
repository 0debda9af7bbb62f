"""K1h (forward with history) and K3 (adjoint) alone, timed with events; checks K3 against the
round-1 arithmetic via an fp64 run on a subset.   python tools/k3_timing.py [cube] [nT]
MRPHY_BWD_VARIANT = waves/SIMD of the K3 build (2, 3, 4)."""
import os
import sys
import torch
sys.path[:0] = ['.', 'oracle']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda', 0)
sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def xcc_report(nb=4096):
    lib = mrphy_amd.require_library()
    out = torch.full((nb,), -1, dtype=torch.int32, device=dev)
    assert lib.mrphy_debug_xcc_map(out.data_ptr(), nb, torch.cuda.current_stream(dev).cuda_stream) == 0
    torch.cuda.synchronize()
    x = out.cpu()
    rr = bool(((x - x[0]) % 8 == (torch.arange(nb) % 8)).all())
    per = [int((x == k).sum()) for k in range(8)]
    return f'xcc of blocks 0..15: {x[:16].tolist()}  round-robin: {rr}  blocks per xcc: {per}'


print(xcc_report(), flush=True)
with torch.no_grad():
    beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
beff.requires_grad_(True)
Mi = sp['M0'].clone().requires_grad_(True)
kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
tf, tb = [], []
for it in range(6):
    a, b, c = ev(), ev(), ev()
    a.record()
    Mo = sims.blochsim(Mi, beff, **kw)
    b.record()
    gMi, gB = torch.autograd.grad(Mo, (Mi, beff), torch.ones_like(Mo))
    c.record()
    torch.cuda.synchronize()
    if it:
        tf.append(a.elapsed_time(b)); tb.append(b.elapsed_time(c))
    if it < 5:
        del gB
ss = n ** 3 * nT
print(f'{n}^3 x {nT} BWD_VARIANT={os.environ.get("MRPHY_BWD_VARIANT", "default")}: K1h {min(tf):.3f} ms = '
      f'{24 * ss / min(tf) / 1e9:.3f} TB/s ({24 * ss / min(tf) / 8e9:.1%}), K3 {min(tb):.3f} ms = '
      f'{36 * ss / min(tb) / 1e9:.3f} TB/s ({36 * ss / min(tb) / 8e9:.1%})', flush=True)
print('  after:', xcc_report(), flush=True)
# accuracy of the adjoint on a subset, against the fp64 kernels
idx = torch.arange(0, n ** 3, max(1, n ** 3 // 2048), device=dev)[:2048]
b64 = beff.detach()[:, idx].double().requires_grad_(True)
M64 = sp['M0'][:, idx].double().requires_grad_(True)
k64 = {k: (v[:, idx] if v.ndim > 1 and v.shape[1] > 1 else v).double() for k, v in kw.items()}
torch.autograd.grad(sims.blochsim(M64, b64, **k64), (M64, b64), torch.ones(1, idx.numel(), 3, device=dev, dtype=torch.float64))
g64 = torch.autograd.grad(sims.blochsim(M64, b64, **k64), (M64, b64), torch.ones(1, idx.numel(), 3, device=dev, dtype=torch.float64))
rel = lambda x, y: float((x.double() - y).norm() / y.norm())  # noqa: E731
print(f'   adjoint vs fp64 kernels on {idx.numel()} spins: grad_Mi {rel(gMi[:, idx], g64[0]):.2e}, grad_Beff {rel(gB[:, idx], g64[1]):.2e}')
