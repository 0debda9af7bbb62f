"""Why is K1 slower right behind K0 on a one-generation grid?  64^3 x 1024 and the 1/8 shard of 128^3 x 4096:
K1 timed (events) (a) repeated alone, (b) right behind K0 on the same buffer, (c) behind K0 and a host sync +
1 ms pause, (d) behind K0 that wrote ANOTHER buffer."""
import json, statistics, sys, time
import torch
sys.path[:0] = ['.']
import mrphy_amd
from mrphy_amd import beffective, sims, synth
dev = torch.device('cuda', 0)
for n, nT, shard in ((64, 1024, 1), (128, 4096, 8)):
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    if shard > 1:
        m = n ** 3 // shard
        sp = {k: (v[:, :m].contiguous() if torch.is_tensor(v) and v.ndim >= 2 and v.shape[1] == n ** 3 else v) for k, v in sp.items()}
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    k0 = lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
    res = {}
    with torch.no_grad():
        beff = k0(); other = k0()
        for mode in ('alone', 'behind_K0', 'behind_K0_sync_pause', 'behind_K0_other_buffer'):
            ts = []
            for it in range(14):
                if mode == 'behind_K0': beff = k0()
                if mode == 'behind_K0_sync_pause': beff = k0(); torch.cuda.synchronize(); time.sleep(0.001)
                if mode == 'behind_K0_other_buffer': other = k0()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); Mo = sims.blochsim(sp['M0'], beff, **kw); b.record(); torch.cuda.synchronize()
                if it >= 2: ts.append(a.elapsed_time(b))
            res[mode] = [round(statistics.median(ts), 4), round(min(ts), 4)]
    print(json.dumps(dict(cube=n, nT=nT, shard_of=shard, K1_ms_median_min=res)), flush=True)
    del beff, other
