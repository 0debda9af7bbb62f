"""Parallel-transmit rfgr2beff, same process (dev build: MRPHY_K0_STEPS, MRPHY_K0_PK), 64^3 x 1024: the
element-per-thread builds of round 2, the step-per-thread kernel (k_rfgr2beff_steps, b1 by LDS broadcast)
and the packed-scalar kernel for exact coil counts (k_rfgr2beff_pk): time per coil count and bitwise
equality of the three outputs.   python tools/k0_mc_ab.py OUT.json"""
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
res = []
for nC in (2, 3, 4, 8, 9, 12, 16, 17, 24, 32):
    rf = (0.05 * torch.randn((1, 2, nT, nC), generator=g)).to(dev)
    b1 = torch.randn((1, n ** 3, 2, nC), generator=g).to(dev)
    out = {}
    row = {'nC': nC}
    for variant in ('elements', 'steps', 'pk') * 2:
        os.environ['MRPHY_K0_STEPS'] = '0' if variant == 'elements' else '1'
        os.environ['MRPHY_K0_PK'] = '1' if variant == 'pk' else '0'
        ts = []
        with torch.no_grad():
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                beff = beffective.rfgr2beff(rf, p['gr'], sp['loc'], Δf=sp['Δf'], b1Map=b1, γ=sp['γ'])
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        key = variant + '_ms'
        row[key] = min(row.get(key, 1e9), round(min(ts[1:]), 4))
        out[variant] = beff
    # (pk serves the exact coil counts 4 / 8 / 12 / 16 / 24 / 32; at any other count the knob changes nothing)
    row['bitwise_equal'] = bool(torch.equal(out['elements'], out['steps']) and torch.equal(out['steps'], out['pk']))
    row['speedup_pk_over_steps'] = round(row['steps_ms'] / row['pk_ms'], 2)
    print(json.dumps(row), flush=True)
    res.append(row)
    del out, beff
os.environ['MRPHY_K0_STEPS'] = '1'; os.environ.pop('MRPHY_K0_PK', None)
json.dump({'workload': '64^3 x 1024 fp32, rfgr2beff with a b1 map', 'runs': res}, open(sys.argv[1], 'w'), indent=1)
