"""Does the KIND of device allocation decide the placement mode of the kernels that stream writes into it?

hipExtMallocWithFlags offers other memory types than hipMalloc's default: fine-grained (0x1), uncached (0x3),
physically contiguous (0x4); and tools/vmm/vmm_alloc.cpp assembles a block from chunks of a given size, mapped in the
order of their creation or SCATTERED over the virtual range by a stride permutation (`vmm <size> seq|perm`).  One fresh
process per repetition; per process R blocks of each type, allocated in rotation; on each block K1h (history = the
block, one part) and K3 (grad_Beff = the block) through the C ABI on raw pointers, a memset, and K1 (read-only) with
Beff IN the block for the read side.

    python tools/alloc_flags_ab.py OUT.json [--procs 4] [--blocks 3] [--n 64] [--nT 2048] [--kinds flags|vmm]
    (child)  python tools/alloc_flags_ab.py --child N NT BLOCKS KINDS
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
FLAGS = (('default (hipMalloc)', None), ('hipDeviceMallocDefault', 0x0), ('finegrained', 0x1), ('uncached', 0x3),
         ('contiguous', 0x4))
MiB = 1 << 20
VMM = (('default (hipMalloc)', None), ('contiguous', 0x4), ('vmm 1024 MiB seq', (1024 * MiB, 0)), ('vmm 1024 MiB perm', (1024 * MiB, 1)),
       ('vmm 128 MiB perm', (128 * MiB, 1)), ('vmm 16 MiB perm', (16 * MiB, 1)), ('vmm 2 MiB perm', (2 * MiB, 1)),
       ('vmm 2 MiB seq', (2 * MiB, 0)))
if os.environ.get('VMM_LIST'):          # e.g. VMM_LIST="2:1,4:1,8:1,16:1" = (MiB : permute) ...
    VMM = (('default (hipMalloc)', None),) + tuple(
        (f"vmm {int(x.split(':')[0])} MiB {('seq', 'perm', 'random')[int(x.split(':')[1])]}", (int(x.split(':')[0]) * MiB, int(x.split(':')[1])))
        for x in os.environ['VMM_LIST'].split(','))
VMM_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vmm', 'libvmm_alloc.so')


def child(n, nT, R, kinds):
    import ctypes
    import torch
    sys.path[:0] = [ROOT]
    import mrphy_amd
    from mrphy_amd import _lib
    dev = torch.device('cuda', 0)
    torch.zeros(1, device=dev)
    lib = mrphy_amd.require_library()
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    code, N, nM = _lib.F32P, 1, n ** 3
    numel = nM * nT * 3
    nbytes = numel * 4
    stream = torch.cuda.current_stream(dev).cuda_stream
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    field = torch.empty(numel, device=dev)
    field.uniform_(-2.0, 2.0)
    hist0 = torch.empty(lib.mrphy_blochsim_hist_bytes(code, N, nM, nT) // 4, device=dev)
    Mi = torch.zeros((N, nM, 3), device=dev)
    Mi[..., 2] = 1
    Mo, gMi, gMo = torch.empty_like(Mi), torch.empty_like(Mi), torch.ones_like(Mi)
    g = torch.tensor(2 * 3.141592653589793 * 4257.6 * 4e-6, device=dev)
    E1, E2 = torch.tensor(0.999996, device=dev), torch.tensor(0.99994, device=dev)
    E1m1 = E1 - 1
    blocks = []
    vl = None
    if kinds == 'vmm':
        vl = ctypes.CDLL(VMM_SO)
        vl.mrphy_vmm_config_chunk_bytes.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t]
        vl.mrphy_vmm_alloc.restype = ctypes.c_void_p
        vl.mrphy_vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for i in range(R):
        for name, fl in (VMM if kinds == 'vmm' else FLAGS):
            p = ctypes.c_void_p()
            if isinstance(fl, tuple):
                vl.mrphy_vmm_config_chunk_bytes(fl[0], fl[1], 1 << 20)
                t0 = time.perf_counter()
                p = ctypes.c_void_p(vl.mrphy_vmm_alloc(nbytes, 0, None))
                rc = 0 if p.value else -1
                if i == 0:
                    print(f'# {name}: built in {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
                if p.value:                       # the mapping is live before anything of ours touches it
                    assert hip.hipMemsetAsync(p, 0, nbytes, stream) == 0
                    torch.cuda.synchronize()
            else:
                rc = hip.hipMalloc(ctypes.byref(p), nbytes) if fl is None else hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, fl)
            if rc != 0 or not p.value:
                print(f'# {name}: allocation failed rc={rc}', flush=True)
                continue
            blocks.append((name, p.value))

    def k1h(ptr):
        tab = (ctypes.c_void_p * 1)(ptr)
        rc = lib.mrphy_blochsim_fwd_parts(code, Mi.data_ptr(), field.data_ptr(), g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                          E2.data_ptr(), 0, 0, E1m1.data_ptr(), Mo.data_ptr(), tab, 1, 0, N, nM, nT, stream)
        assert rc == 0, rc

    def k3(ptr):
        tab = (ctypes.c_void_p * 1)(hist0.data_ptr())
        rc = lib.mrphy_blochsim_bwd_parts(code, tab, 1, 0, field.data_ptr(), g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                          E2.data_ptr(), 0, 0, gMo.data_ptr(), gMi.data_ptr(), ptr, None, N, nM, nT, stream)
        assert rc == 0, rc

    def k1_reading(ptr):          # Beff IN the block: the read side
        rc = lib.mrphy_blochsim_fwd_parts(code, Mi.data_ptr(), ptr, g.data_ptr(), 0, 0, E1.data_ptr(), 0, 0,
                                          E2.data_ptr(), 0, 0, E1m1.data_ptr(), Mo.data_ptr(), None, 0, 0, N, nM, nT, stream)
        assert rc == 0, rc

    def timed(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            a, b = ev(), ev()
            a.record(); fn(); b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return round(min(ts), 4)

    k1h(hist0.data_ptr())                     # the history K3 reads
    torch.cuda.synchronize()
    rows = []
    for name, ptr in blocks:
        t1h = timed(lambda: k1h(ptr))
        t3 = timed(lambda: k3(ptr))           # (grad_Beff values are then in the block: finite numbers, read below as a field)
        tf = timed(lambda: hip.hipMemsetAsync(ptr, 0, nbytes, stream))
        t1 = timed(lambda: k1_reading(ptr))
        rows.append(dict(kind=name, K1h_ms=t1h, K3_ms=t3, memset_ms=tf, K1_read_ms=t1))
    print('RESULT ' + json.dumps(rows), flush=True)
    torch.cuda.synchronize()
    sys.stdout.flush()
    os._exit(0)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(*[int(x) for x in sys.argv[2:5]], sys.argv[5] if len(sys.argv) > 5 else 'flags')
        sys.exit(0)
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--procs', type=int, default=4)
    ap.add_argument('--blocks', type=int, default=3)
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--nT', type=int, default=2048)
    ap.add_argument('--kinds', default='flags', choices=['flags', 'vmm'])
    a = ap.parse_args()
    allrows = []
    for pr in range(a.procs):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(a.n), str(a.nT), str(a.blocks), a.kinds],
                           capture_output=True, text=True, timeout=250)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
        for ln in r.stdout.splitlines():
            if ln.startswith('#'):
                print(ln, flush=True)
        if r.returncode or not line:
            print('child failed', r.returncode, r.stdout[-2000:], r.stderr[-1500:], flush=True)
            sys.exit(1)                      # no further GPU step after a failed one
        rows = json.loads(line[0][7:])
        for x in rows:
            x['process'] = pr
        allrows += rows
    summ = {}
    for kind in [f[0] for f in (VMM if a.kinds == 'vmm' else FLAGS)]:
        rs = [x for x in allrows if x['kind'] == kind]
        if rs:
            summ[kind] = {k: sorted(x[k] for x in rs) for k in ('K1h_ms', 'K3_ms', 'memset_ms', 'K1_read_ms')}
    json.dump(dict(cube=a.n, nT=a.nT, summary=summ, rows=allrows), open(a.out, 'w'), indent=1)
    for k, v in summ.items():
        print(k, json.dumps(v), flush=True)
