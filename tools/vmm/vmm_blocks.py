"""Experiment (round 6): are blocks assembled from several physical allocations (tools/vmm/vmm_alloc.cpp: hipMemCreate x CHUNKS,
mapped contiguously) of the fast placement kind for the kernels that stream writes into them?

One fresh process per repetition (the kind of an allocation is the process's lottery): R plain blocks from the caching
allocator and R chunked blocks from a torch MemPool over the pluggable allocator, allocated alternately; on each block
K0 (rfgr2beff(out=block)), K1h (history = block, one part), K3 (grad_Beff = block) and fill_ are timed; results compared
bit for bit between the two kinds.

    python tools/vmm/vmm_blocks.py OUT.json [--procs 4] [--n 64] [--nT 2048] [--chunks 4] [--blocks 4]
    (child)  python tools/vmm/vmm_blocks.py --child N NT CHUNKS BLOCKS
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
SO = os.path.join(HERE, 'libvmm_alloc.so')


def child(n, nT, chunks, R):
    import ctypes
    import torch
    sys.path[:0] = [ROOT]
    import mrphy_amd  # noqa: F401
    from mrphy_amd import beffective, sims, synth
    from mrphy_amd.workspace import _Pair
    dev = torch.device('cuda', 0)
    vl = ctypes.CDLL(SO)
    vl.mrphy_vmm_config.argtypes = [ctypes.c_int, ctypes.c_size_t]
    vl.mrphy_vmm_config(chunks, 64 << 20)
    alloc = torch.cuda.memory.CUDAPluggableAllocator(SO, 'mrphy_vmm_alloc', 'mrphy_vmm_free')
    pool = torch.cuda.MemPool(alloc.allocator())
    nM = n ** 3
    numel = nM * nT * 3
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev, seed_M0=4)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    kw = dict(T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    field = torch.empty(numel, device=dev)
    other = torch.empty(numel, device=dev)
    blocks = []
    for i in range(R):
        blocks.append(('plain', torch.empty(numel, device=dev)))
        with torch.cuda.use_mem_pool(pool):
            b = torch.empty(numel, device=dev)
        b.zero_()
        torch.cuda.synchronize()                 # the mapping is live before anything of ours touches it
        blocks.append((f'chunked x{chunks}', b))

    def timed(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            a, b = ev(), ev()
            a.record(); fn(); b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return round(min(ts), 4)

    with torch.no_grad():
        beff0 = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=field.view(1, nM, nT, 3))
    beff = field.view(1, nM, nT, 3).requires_grad_(True)
    gMo = torch.ones((1, nM, 3), device=dev)
    rows, ref = [], {}
    for kind, blk in blocks:
        with torch.no_grad():
            out = blk.view(1, nM, nT, 3)
            t0 = timed(lambda: beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=out))
            same_k0 = bool(torch.equal(out, beff0))
        t1h = timed(lambda: sims.blochsim(sp['M0'], beff, workspace=_Pair(blk, other), **kw))
        Mo = sims.blochsim(sp['M0'], beff, workspace=_Pair(other, blk), **kw)
        t3 = timed(lambda: torch.autograd.grad(Mo, beff, gMo, retain_graph=True))
        gsum = float(blk[:numel].double().sum())
        tf = timed(lambda: blk.fill_(0.5))
        ref.setdefault('g', gsum)
        rows.append(dict(kind=kind, ptr=hex(blk.data_ptr()), K0_ms=t0, K1h_ms=t1h, K3_ms=t3, fill_ms=tf,
                         K0_same_bits=same_k0, K3_same_sum=gsum == ref['g']))
        del Mo
    print('RESULT ' + json.dumps(rows), flush=True)
    torch.cuda.synchronize()
    # (leave without running destructors: at interpreter shutdown the pool would call the pluggable allocator's free after
    # the HIP runtime has begun to unload -- the first run of this tool ended in SIGSEGV there, after its results were out)
    sys.stdout.flush()
    os._exit(0)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(*[int(x) for x in sys.argv[2:6]])
        sys.exit(0)
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--procs', type=int, default=4)
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--nT', type=int, default=2048)
    ap.add_argument('--chunks', type=int, default=4)
    ap.add_argument('--blocks', type=int, default=4)
    a = ap.parse_args()
    allrows = []
    for pr in range(a.procs):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(a.n), str(a.nT), str(a.chunks),
                            str(a.blocks)], capture_output=True, text=True, timeout=250)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
        if r.returncode or not line:
            print('child failed', r.returncode, r.stdout[-3000:], r.stderr[-1500:], flush=True)
            sys.exit(1)                      # no further GPU step after a failed one
        rows = json.loads(line[0][7:])
        for x in rows:
            x['process'] = pr
            print(json.dumps(x), flush=True)
        allrows += rows
    summ = {}
    for kind in sorted(set(x['kind'] for x in allrows)):
        rs = [x for x in allrows if x['kind'] == kind]
        summ[kind] = {k: sorted(x[k] for x in rs) for k in ('K0_ms', 'K1h_ms', 'K3_ms', 'fill_ms')}
        summ[kind]['same_bits'] = all(x['K0_same_bits'] and x['K3_same_sum'] for x in rs)
    json.dump(dict(cube=a.n, nT=a.nT, chunks=a.chunks, summary=summ, rows=allrows), open(a.out, 'w'), indent=1)
    for k, v in summ.items():
        print(k, json.dumps(v), flush=True)
