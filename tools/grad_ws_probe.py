"""Does a placement-probed workspace put the gradient route's write kernels in their fast mode?

For each workload: the materialised gradient iteration (rfgr2beff -> blochsim with history -> backward) through the
caching allocator (what rounds 1-4 measured) and through `mrphy_amd.workspace.GradWorkspace`, K1h and K3 launch
durations by HIP events around the C-ABI calls, gradients compared bit for bit.

    python tools/grad_ws_probe.py OUT.json [candidates]
"""
import json
import sys

import torch

sys.path[:0] = ['.']
import mrphy_amd  # noqa: E402
from mrphy_amd import beffective, sims, synth, workspace  # noqa: E402

dev = torch.device('cuda', 0)
cands = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
lib = mrphy_amd.require_library()
launches = {}


def hook(name):
    fn = getattr(lib, name)

    def call(*a):
        e0, e1 = ev(), ev()
        e0.record(); rc = fn(*a); e1.record()
        launches.setdefault(name, []).append((e0, e1))
        return rc
    setattr(lib, name, call)
    return fn


orig = {n: hook(n) for n in ('mrphy_blochsim_fwd_parts', 'mrphy_blochsim_bwd_parts')}


def iterate(sp, p, ws, iters):
    launches.clear()
    g = None
    for _ in range(iters):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=None if ws is None else ws.beff)
        Mo = sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'], workspace=ws)
        Mo.sum().backward()
        g = (rf.grad.clone(), gr.grad.clone(), Mo.detach().clone())
        del beff, Mo
    torch.cuda.synchronize()
    ms = {k: [a.elapsed_time(b) for a, b in v][1:] for k, v in launches.items()}
    return g, {k: (round(min(v), 4), round(sum(v) / len(v), 4)) for k, v in ms.items()}


res = []
for n, nT, dtype in ((64, 2048, torch.float32), (128, 1024, torch.float32), (64, 1024, torch.float64)):
    sp = synth.cube_spins(n, dtype=dtype, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=dtype, device=dev)
    ss = n ** 3 * nT
    esz = 8 if dtype == torch.float64 else 4
    b_h, b_3 = 6 * esz * ss, 9 * esz * ss            # K1h: Beff read + history written; K3: 2 reads + 1 write
    torch.cuda.empty_cache()
    g0, t0 = iterate(sp, p, None, 6)
    ws = workspace.GradWorkspace((1, n ** 3, nT, 3), dtype, dev, candidates=cands)
    g1, t1 = iterate(sp, p, ws, 6)
    same = all(bool((a == b).all()) for a, b in zip(g0, g1))
    fr = lambda t, b: round(b / (t * 1e-3) / 8e12, 3)  # noqa: E731
    r = dict(cube=n, nT=nT, dtype=str(dtype), candidates=cands, workspace=ws.report,
             allocator=dict(K1h_ms_min_avg=t0['mrphy_blochsim_fwd_parts'], K3_ms_min_avg=t0['mrphy_blochsim_bwd_parts'],
                            K1h_frac=fr(t0['mrphy_blochsim_fwd_parts'][1], b_h), K3_frac=fr(t0['mrphy_blochsim_bwd_parts'][1], b_3)),
             probed=dict(K1h_ms_min_avg=t1['mrphy_blochsim_fwd_parts'], K3_ms_min_avg=t1['mrphy_blochsim_bwd_parts'],
                         K1h_frac=fr(t1['mrphy_blochsim_fwd_parts'][1], b_h), K3_frac=fr(t1['mrphy_blochsim_bwd_parts'][1], b_3)),
             gradients_and_Mo_bit_identical=same, reserved_GB=round(torch.cuda.memory_reserved() / 1e9, 2))
    print(json.dumps(r), flush=True)
    res.append(r)
    del ws, g0, g1, sp, p
    torch.cuda.empty_cache()
json.dump({'note': 'materialised gradient iteration, K1h / K3 launch ms (min, mean over 5) through the caching allocator '
                   'and through a placement-probed GradWorkspace; fractions of 8 TB/s on 24 / 36 B per spin-step '
                   '(48 / 72 in fp64)', 'runs': res}, open(sys.argv[1], 'w'), indent=1)
