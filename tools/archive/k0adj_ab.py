"""Parallel-transmit adjoint of rfgr2beff (grad_Beff -> grad_rf, grad_gr), same process, 64^3 x 1024, per
coil count: the step-per-thread pass with the row's coefficients in SGPRs (k_rfgr2beff_bwd_sgpr, the
shipped default), its DPP predecessor (k_rfgr2beff_bwd_steps, TP = 1 / 2 time points per thread) and round
2's element-per-thread pass (dev build knob MRPHY_K0ADJ_TP = 1 / 2 / 0); time, and each variant's distance to
the same sums formed in fp64.   python tools/k0adj_ab.py OUT.json"""
import json
import os
import sys
import torch
sys.path[:0] = ['.', 'tools']
import build_dev  # noqa: E402
build_dev.use()
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import beffective, synth  # noqa: E402
dev = torch.device('cuda', 0)
n, nT = 64, 1024
sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
g = torch.Generator(device='cpu').manual_seed(5)
gB = torch.randn((1, n ** 3, nT, 3), generator=g).to(dev)
res = []


def exact(b1):
    r"""grad_rf, grad_gr in fp64 (chunked over spins)."""
    nC = b1.shape[-1]
    grf = torch.zeros((1, 2, nT, nC), dtype=torch.float64, device=dev)
    ggr = torch.zeros((1, 3, nT), dtype=torch.float64, device=dev)
    for s in range(0, n ** 3, 32768):
        G = gB[0, s:s + 32768].double()
        br, bi = b1[0, s:s + 32768, 0].double(), b1[0, s:s + 32768, 1].double()
        grf[0, 0] += G[..., 0].T @ br + G[..., 1].T @ bi
        grf[0, 1] += G[..., 1].T @ br - G[..., 0].T @ bi
        ggr[0] += sp['loc'][0, s:s + 32768].double().T @ G[..., 2]
    return grf, ggr


rel = lambda a, b: float((a.double() - b).norm() / b.norm())  # noqa: E731
for nC in (2, 4, 8, 9, 12, 16, 17, 24, 32):
    rf = (0.05 * torch.randn((1, 2, nT, nC), generator=g)).to(dev).requires_grad_(True)
    gr = p['gr'].clone().requires_grad_(True)
    b1 = torch.randn((1, n ** 3, 2, nC), generator=g).to(dev)
    beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
    want = exact(b1)
    row = {'nC': nC}
    for rep in range(2):
        for tp in (0, 1, 12, -1):                    # -1: the default, coefficients in SGPRs
            if tp < 0:
                os.environ.pop('MRPHY_K0ADJ_TP', None)
            else:
                os.environ['MRPHY_K0ADJ_TP'] = str(tp)
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                grf, ggr = torch.autograd.grad(beff, (rf, gr), gB, retain_graph=True)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            name = {0: 'elements', 1: 'dpp_tp1', 12: 'sgpr_tp2', -1: 'sgpr'}[tp]
            key = name + '_ms'
            row[key] = min(row.get(key, 1e9), round(min(ts[1:]), 4))
            row[name + '_rel_rf_gr'] = [float(f'{rel(grf, want[0]):.2e}'),
                                                                    float(f'{rel(ggr, want[1]):.2e}')]
    print(json.dumps(row), flush=True)
    res.append(row)
    del beff
os.environ.pop('MRPHY_K0ADJ_TP', None)
json.dump({'workload': '64^3 x 1024 fp32, adjoint of rfgr2beff with a b1 map (grad_Beff random normal)', 'runs': res},
          open(sys.argv[1], 'w'), indent=1)
