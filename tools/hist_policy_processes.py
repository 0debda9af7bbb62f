"""The reference-signature gradient route (rfgr2beff -> sims.blochsim -> backward) in FRESH processes, one per
(history policy, repetition): what K1h / K3 / K0 run at when nothing is probed and the blocks are whatever the caching
allocator draws in a new process.  VERDICT r5 item 3's criterion: "fresh processes, no probing, K1h <= 2.30 ms at
64^3 x 2048 in >= 5 of 6 processes".

    python tools/hist_policy_processes.py OUT.json [--n 64] [--nT 2048] [--procs 6] [--f64]
                                          [--variants parts1,parts2,parts4,parts8]
    (child)  python tools/hist_policy_processes.py --child VARIANT N NT DTYPE
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def child(variant, n, nT, f64):
    import torch
    sys.path[:0] = [ROOT]
    import mrphy_amd
    from mrphy_amd import beffective, sims, synth, _hist
    dev = torch.device('cuda', 0)
    dt_ = torch.float64 if f64 else torch.float32
    if variant == 'parts1':
        _hist.set_policy(parts=1)
    else:
        _hist.set_policy(parts=int(variant[5:]))
    lib = mrphy_amd.require_library()
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    launches = {'mrphy_blochsim_fwd_parts': [], 'mrphy_blochsim_bwd_parts': [], 'mrphy_rfgr2beff_st': []}

    def hook(name):
        fn = getattr(lib, name)

        def call(*args):
            e0, e1 = ev(), ev()
            e0.record(); rc = fn(*args); e1.record()
            launches[name].append((e0, e1))
            return rc
        setattr(lib, name, call)
    for nm in launches:
        hook(nm)
    sp = synth.cube_spins(n, dtype=dt_, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=dt_, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    first_fwd_host_ms = None
    W, K = 1, 4
    for it in range(W + K):
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = beffective.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Mo = sims.blochsim(sp['M0'], beff, **kw)
        torch.cuda.synchronize()
        if it == 0:
            first_fwd_host_ms = (time.perf_counter() - t0) * 1e3
        Mo.sum().backward()
        del beff, Mo
    torch.cuda.synchronize()

    def ms(name):
        v = sorted(a.elapsed_time(b) for a, b in launches[name][W:])
        return [round(v[0], 4), round(v[len(v) // 2], 4)]
    print('RESULT ' + json.dumps(dict(variant=variant, K0=ms('mrphy_rfgr2beff_st'), K1h=ms('mrphy_blochsim_fwd_parts'),
                                      K3=ms('mrphy_blochsim_bwd_parts'), first_forward_host_ms=round(first_fwd_host_ms, 1),
                                      reserved_GB=round(torch.cuda.memory_reserved() / 1e9, 2))), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == 'f64')
        sys.exit(0)
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--nT', type=int, default=2048)
    ap.add_argument('--procs', type=int, default=6)
    ap.add_argument('--f64', action='store_true')
    ap.add_argument('--variants', default='parts1,parts2,parts4,parts8')
    a = ap.parse_args()
    rows = []
    for rep in range(a.procs):
        for v in a.variants.split(','):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', v, str(a.n), str(a.nT),
                                'f64' if a.f64 else 'f32'], capture_output=True, text=True, timeout=280)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
            if r.returncode or not line:
                print('child failed', v, r.returncode, r.stderr[-600:], flush=True)
                sys.exit(1)          # no further GPU step after a failed one
            row = json.loads(line[0][7:])
            row['process'] = rep
            rows.append(row)
            print(json.dumps(row), flush=True)
    es = 8 if a.f64 else 4
    b1h, b3 = 24 * a.n ** 3 * a.nT * es // 4, 36 * a.n ** 3 * a.nT * es // 4
    summary = {}
    for v in a.variants.split(','):
        rs = [r for r in rows if r['variant'] == v]
        k1h, k3 = [r['K1h'][1] for r in rs], [r['K3'][1] for r in rs]
        summary[v] = dict(K1h_ms_median_per_process=k1h, K3_ms_median_per_process=k3,
                          K1h_frac_hbm=[round(b1h / (t * 1e-3) / 8e12, 3) for t in k1h],
                          K3_frac_hbm=[round(b3 / (t * 1e-3) / 8e12, 3) for t in k3],
                          first_forward_host_ms=[r['first_forward_host_ms'] for r in rs])
    json.dump(dict(cube=a.n, nT=a.nT, dtype='f64' if a.f64 else 'f32', processes=a.procs, summary=summary, rows=rows),
              open(a.out, 'w'), indent=1)
    for v, s in summary.items():
        print(v, 'K1h', s['K1h_ms_median_per_process'], 'K3', s['K3_ms_median_per_process'], flush=True)
