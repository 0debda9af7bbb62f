import os, sys, time, torch
sys.path.insert(0, '.')
import torch.distributed as dist
import mrphy_amd
from mrphy_amd import beffective, sims, synth
use_dist = 'RANK' in os.environ
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
if use_dist:
    dist.init_process_group('nccl', device_id=dev)
print('env', {k: v for k, v in os.environ.items() if 'ALLOC' in k or 'NCCL' in k or 'HSA' in k or 'TORCH' in k}, flush=True)
n, nT = 128, 4096
sp = synth.cube_spins(n, dtype=torch.float32, device=dev); p = synth.pulse(nT, dtype=torch.float32, device=dev)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        Mo = sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del beff
        if use_dist:
            from mrphy_amd.dist import all_gather_spins
            Mg = all_gather_spins(Mo, n ** 3, force=True)
            torch.cuda.synchronize()
    t3 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f'it {it} (gather in loop): K0 leg {1e3*(t1-t0):.1f} ms, K1 leg {1e3*(t2-t1):.1f} ms, barrier {1e3*(t3-t2):.1f} ms; device mallocs {st.get("num_device_alloc")}, frees {st.get("num_device_free")}, reserved {st["reserved_bytes.all.current"]/1e9:.1f} GB', flush=True)
if use_dist:
    dist.destroy_process_group()
