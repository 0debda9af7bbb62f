#!/usr/bin/env python3
r"""Benchmark of the Bloch-simulation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the synthetic workload: ``rfgr2beff`` (K0, writes
Beff) followed by ``sims.blochsim`` (K1, integrates it), through the drop-in Python API and the
C ABI -- plus, for N > 1, the RCCL all-gather of the final magnetisation.  Inputs are resident
in HBM before the timed region.  Metric (BASELINE.json): spin-steps/s = spins x nT x K / time.

At N = 1 the step runs through exactly the reference's signatures (a fresh Beff per step); the same step with a
placement-probed block (``rfgr2beff(..., out=, store=)``, mrphy_amd.workspace.BeffArena) is reported beside it
(``arena_step``) and is what the ranks of an N > 1 run use.

Workload at N = 1: BASELINE.json configs[2], the 128^3 cube (2 097 152 spins) x 4096 steps,
fp32, closed-form synthetic inputs (mrphy_amd/synth.py).  For N > 1 the same
cube is sharded over the ranks (configs[3]: total work fixed => "scaling": "strong").

Besides the contract fields the JSON line carries
  roofline      of the dominant kernel K1 (HBM-bound, 12 B/spin-step algorithmic), its launch
                duration measured live with HIP events on the launch stream;
  kernels       K0 / K1 / fused-K2 durations and rates;
  cpu_baseline  the oracle's op-for-op restatement of the reference's CPU PyTorch path, timed
                on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9     # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 fp32 lane-ops/s


def newest_profile(suffix):
    r"""The newest round's ``profiles/rNN_<suffix>`` (the committed rocprofv3 summaries are named per round); a path
    that does not exist when there is none."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{suffix}')))
    return hits[-1] if hits else os.path.join(ROOT, 'profiles', f'r00_{suffix}')


def rel(path):
    return os.path.relpath(path, ROOT)


K2_PMC = newest_profile('k2_pmc.json')
K2B_PMC = newest_profile('k2b_pmc.json')
TRAFFIC = newest_profile('traffic.json')


def k2_valu_profile(mode, path=None):
    r"""VALU instructions per wave-step of the fused kernel K2 (or, with `path`, K2b) in `mode` ('precise' |
    'fast'), from the rocprofv3 PMC passes of this tree (profiles/rNN_k2_pmc.json, written by
    tools/k2_pmc_profile.py from SQ_INSTS_VALU and its per-type breakdown): (all instructions, those that
    issue at half rate -- fp64 FMAs and fp32<->fp64 conversions).  None -- nothing is assumed -- when the file
    is missing OR was collected on other kernel sources than the ones this library was built from (the file
    records `source_id`, SHA-1 over mrphy.py_amd/csrc): a stale instruction count is not reported as measured."""
    try:
        j = json.load(open(path or K2_PMC))
        if j.get('source_id') != source_id():
            return None
        e = j['modes'][mode]
        return float(e['valu_insts_per_wave_step']), float(e['half_rate_insts_per_wave_step'])
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--cube', '--n', dest='n', type=int, default=None,
                    help='cube edge (default 128 = BASELINE configs[2]; 64 = configs[4] with --mode '
                         'grad); use --cube under torchrun, whose own parser treats --n as an '
                         'abbreviation')
    ap.add_argument('--nT', type=int, default=None, help='default 4096 (2048 with --mode grad)')
    ap.add_argument('--config', type=int, default=None, choices=[1, 2, 4],
                    help='BASELINE.json configs[i]: 1 = 64^3 x 1024, 2 = 128^3 x 4096 (the default), '
                         '4 = 64^3 x 2048 forward + backward (= --mode grad); sets --cube / --nT / --mode')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-spins', type=int, default=32768,
                    help='spins per cpu_baseline chunk (x all nT steps; SURVEY 8d: 32 768); the '
                         'chunks stop early once --cpu-budget seconds are used')
    ap.add_argument('--cpu-budget', type=float, default=150.0,
                    help='cpu_baseline: stop starting new chunks after this many seconds')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--verify-full', action='store_true',
                    help='N=1: compare Mo of ALL spins with oracle/bloch_c.c (fp64 arithmetic on the '
                         'same fp32 inputs and constants; ~30 s on 16 cores at 128^3 x 4096)')
    ap.add_argument('--mode', default='fwd', choices=['fwd', 'grad'],
                    help="'grad': BASELINE configs[4] -- 64^3 x 2048, coarse pulse -> interpT -> "
                         "simulate -> backward to the coarse pulse (not the contract line; prints "
                         "per-stage times)")
    ap.add_argument('--fused-only', action='store_true',
                    help='grad mode: skip the materialised route (at 128^3 x 4096 it would need '
                         'Beff + history + grad_Beff = 309 GB)')
    ap.add_argument('--no-interp', action='store_true',
                    help='grad mode: differentiate w.r.t. the fine pulse, no interpT stage')
    ap.add_argument('--dry-launch', action='store_true',
                    help='with --gpus N: start the N rank processes, have each print its '
                         'RANK/LOCAL_RANK/WORLD_SIZE as one JSON line and exit (no GPU call): checks '
                         'the launcher on a CPU-only machine')
    ap.add_argument('--shard-of', type=int, default=0, metavar='S',
                    help='N=1 only: after the normal run, rehearse ONE rank of an S-GPU run on this GPU -- '
                         "rank 0's 1/S block of the cube, RCCL initialised at world size 1, the "
                         'asynchronous all-gather taken -- and add shard_rehearsal{..., expected_speedup = '
                         't_full / t_shard} to the JSON line (an estimate for S GPUs, not a measurement)')
    ap.add_argument('--arena', type=int, default=None, metavar='C',
                    help='candidate Beff blocks the placement-aware arena tries before the timed region '
                         '(mrphy_amd.workspace.BeffArena: the step is timed on each, the fastest is kept and '
                         'passed as out= to every rfgr2beff; as many as fit in memory); 0 = the plain reference '
                         'signatures, a fresh Beff per step from the caching allocator.  Default: 0 on one GPU (the '
                         'headline is the reference API; the arena step is reported beside it), 3 per rank for N > 1 '
                         '(the slowest rank sets the time, and one unlucky block in eight is likely)')
    ap.add_argument('--grad-candidates', type=int, default=-1, metavar='C',
                    help='configs[4]: candidate blocks mrphy_amd.workspace.GradWorkspace may draw for grad_Beff (and a '
                         'one-block history) of the materialised gradient route, timed with K1h / K3 before the timed '
                         'iterations; default -1 = the workspace\'s own caps (8 blocks, 32 GiB or four blocks alive, 2 s); '
                         '0 = the caching allocator only')
    ap.add_argument('--grad-route', default='both', choices=['both', 'allocator', 'workspace'],
                    help='configs[4], materialised route: time it with the caching allocator\'s blocks, through the '
                         'placement-probed GradWorkspace, or both (default; the JSON line labels each)')
    ap.add_argument('--collectives', default='torch', choices=['torch', 'c-abi'],
                    help='N > 1 (or one rank under torch.distributed.run): the all-gather of Mo through torch.distributed '
                         '(backend nccl = RCCL; default) or through the C ABI of libmrphy_comm.so (mrphy_comm_allgather_spins: '
                         'RCCL called directly on the compute stream; the communicator\'s id travels over the process group)')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='N=1, configs[2] run: do not also time BASELINE configs[1] and configs[4] after the headline '
                         '(the `configs` object of the JSON line)')
    ap.add_argument('--cpu-chunks', type=int, default=3,
                    help='cpu_baseline: number of spin chunks timed (SURVEY 8d: >= 3, extrapolated)')
    a = ap.parse_args()
    if a.config is not None:
        a.n, a.nT, a.mode = {1: (64, 1024, 'fwd'), 2: (128, 4096, 'fwd'), 4: (64, 2048, 'grad')}[a.config]
    if a.shard_of:
        # the rehearsal brings up a one-rank RCCL group, and every step of the run then sends its result through the
        # (asynchronous) all-gather: at world size 1 RCCL turns that into a device copy, which at configs[1]'s step length
        # (1.1 ms) sits inside the next step's rfgr2beff interval in un-profiled runs (round 6: K0 1.8-2.0 ms instead of
        # 0.55 -- not under rocprofv3, not at the shard's or the headline's step length).  The other configs are the
        # driver's command's business (no process group there): not timed in a rehearsal run.
        a.no_extra_configs = True
    if a.n is None:
        a.n = 64 if a.mode == 'grad' else 128
    if a.nT is None:
        a.nT = 2048 if a.mode == 'grad' else 4096
    return a


def baseline_config(n, nT, mode, world):
    r"""Which entry of BASELINE.json `configs` a run is, from its size -- or that it is none of them."""
    if mode == 'grad':
        return 'BASELINE.json configs[4]' if (n, nT) == (64, 2048) else f'none of BASELINE.json configs ({n}^3 x {nT}, forward + backward)'
    if (n, nT) == (128, 4096):
        return 'BASELINE.json configs[2]' if world == 1 else 'BASELINE.json configs[3]'
    if (n, nT, world) == (64, 1024, 1):
        return 'BASELINE.json configs[1]'
    return f'none of BASELINE.json configs ({n}^3 x {nT}, {world} GPU(s))'


def source_id():
    r"""Identifier of the kernel sources the loaded library was built from (SHA-1 over mrphy.py_amd/csrc with
    comments and blank lines removed, so that an edit of the prose does not orphan a measurement, plus the
    compile flags): the committed
    PMC instruction counts carry it, and are only used when it matches."""
    import hashlib
    import re
    d = os.path.join(ROOT, 'mrphy.py_amd', 'csrc')
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp', '.h')):
            t = open(os.path.join(d, f), encoding='utf-8').read()
            t = re.sub(r'/\*.*?\*/', '', t, flags=re.S)
            t = re.sub(r'//[^\n]*', '', t)
            t = '\n'.join(ln.rstrip() for ln in t.split('\n') if ln.strip())
            h.update(f.encode()); h.update(t.encode())
    from mrphy_amd import _lib                       # ... and the flags they are compiled with
    h.update(repr((_lib.hipcc_flags()[:-1], sorted((str(k), tuple(v)) for k, v in _lib.UNIT_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def host_cores():
    r"""CPU cores this process may really use: affinity mask and cgroup quota, not the host's
    core count (a 1-GPU box exposes a 16-core share of a much larger host)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            c = min(c, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    # MRPHY_BENCH_CORES only LOWERS the count (e.g. to leave cores to a profiler); there is no cap
    return max(1, min(c, int(os.environ.get('MRPHY_BENCH_CORES', c))))


def log(msg):
    print(f'[bench +{time.perf_counter() - T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


# The contract: rank 0 prints ONE JSON line on stdout.  Native libraries do not know that -- RCCL
# writes a five-line banner ("RCCL version : ...", "HIP version : ...") to STDOUT, through C stdio,
# when the process group is initialised (seen in the world-size-1 rehearsal of the rank path on the
# MI355X box), which sys.stdout tricks cannot catch.  So, before anything touches the GPU or
# torch.distributed: keep a duplicate of the real stdout aside and point file descriptor 1 at
# stderr; the JSON line -- and nothing else -- is written to the duplicate.
_REAL_STDOUT = None


def protect_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)


def emit(obj):
    f = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    f.write(json.dumps(obj) + '\n')
    f.flush()


T0 = time.perf_counter()


def cpu_baseline(n, nT, spins, chunks=3, budget_s=150.0):
    r"""The reference's CPU PyTorch path (oracle restatement, same ATen sequence) on a bounded
    sample, as SURVEY 8(d) prescribes for a workload whose Beff cannot be materialised on the host:
    `chunks` chunks of `spins` spins of the same cube x all nT steps (spins are independent),
    rfgr2beff + blochsim on all host cores, each chunk timed; the rate is total spin-steps over total
    time (= the linear extrapolation to the whole cube).  Stops early once `budget_s` is used."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bloch_oracle as O
    from mrphy_amd import synth
    torch.set_num_threads(host_cores())
    idx = synth.subset_indices(n, spins * chunks, seed=99)
    p = synth.pulse(nT, dtype=torch.float32)
    times, Mos = [], []
    for c in range(chunks):
        sp = synth.cube_spins(n, idx[c * spins:(c + 1) * spins], dtype=torch.float32)
        t0 = time.perf_counter()
        with torch.no_grad():
            beff = O.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
            Mos.append(O.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt']))
        times.append(time.perf_counter() - t0)
        del beff
        if sum(times) > budget_s:
            break
    done = len(times)
    dt = sum(times)
    Mo = torch.cat(Mos, dim=1)
    return dict(value=done * spins * nT / dt, unit='spin-steps/s', cores=torch.get_num_threads(),
                kind='port', seconds=round(dt, 2), chunk_seconds=[round(t, 2) for t in times],
                whole_workload_extrapolated_s=round(dt * (n ** 3) / (done * spins), 1),
                fidelity={'oracle_over_reference_cpu_time': {'32768 spins x 256 steps': 0.887, '32768 x 1024': 0.989},
                          'source': 'tests/golden/make_golden.py --check in the build container (round 6): the port is the '
                                    'oracle, results bit-identical to the imported reference, CPU time on these chunk shapes '
                                    'within +-20 % of it (medians of three alternating runs, 8 threads; asserted there)'},
                sample=f'{done} chunks x {spins} spins (seeded subset of the {n}^3 cube) x {nT} steps, '
                       f'fp32, rfgr2beff+blochsim, torch {torch.__version__} CPU; rate = total '
                       f'spin-steps / total time (linear extrapolation over spins)'), Mo, idx[:done * spins]


def cpu_baseline_grad(n, nT, spins, chunks=3, budget_s=60.0):
    r"""configs[4]'s CPU leg: the reference's CPU PyTorch path (oracle restatement: rfgr2beff + the explicit
    BlochSim forward and backward) differentiating sum(Mo) w.r.t. the fine pulse, on `chunks` chunks of `spins`
    seeded spins of the cube x all nT steps; total spin-steps over total time."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bloch_oracle as O
    from mrphy_amd import synth
    torch.set_num_threads(host_cores())
    idx = synth.subset_indices(n, spins * chunks, seed=99)
    p = synth.pulse(nT, dtype=torch.float32)
    times, g_rf, g_gr = [], 0., 0.
    for c in range(chunks):
        sp = synth.cube_spins(n, idx[c * spins:(c + 1) * spins], dtype=torch.float32)
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        t0 = time.perf_counter()
        beff = O.rfgr2beff(rf, gr, sp['loc'], Δf=sp['Δf'], γ=sp['γ'])
        Mo = O.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
        Mo.sum().backward()
        times.append(time.perf_counter() - t0)
        g_rf, g_gr = g_rf + rf.grad, g_gr + gr.grad
        del beff, Mo
        if sum(times) > budget_s:
            break
    done, dt = len(times), sum(times)
    return dict(value=done * spins * nT / dt, unit='spin-steps/s (forward + backward)', cores=torch.get_num_threads(),
                kind='port', seconds=round(dt, 2), chunk_seconds=[round(t, 2) for t in times],
                sample=f'{done} chunks x {spins} spins (seeded subset of the {n}^3 cube) x {nT} steps, fp32, '
                       f'rfgr2beff + blochsim forward + backward of sum(Mo) to the fine rf/gr (no interpT: the '
                       f'reference resamples with scipy on the host, outside autograd), torch {torch.__version__} CPU'), \
        (g_rf, g_gr), idx[:done * spins]


def grad_measure(n, nT, K, W, *, multi=True, fused_only=False, candidates=-1, log_=None, route='both'):
    r"""BASELINE configs[4]: multi-scale pulse design step on one GPU.  A coarse pulse (nT/2 samples
    at 2 dt) is resampled to nT samples with the differentiable on-device ``interpT``, simulated,
    and ``sum(Mo)`` is differentiated back to the coarse ``rf``/``gr`` -- through the materialised route
    (rfgr2beff -> blochsim with history -> adjoints: the reference's own ``sims.blochsim(...).backward()``
    signature), once with every block drawn from the caching allocator and once with the history, ``grad_Beff`` and
    ``Beff`` blocks of a placement-probed ``mrphy_amd.workspace.GradWorkspace`` (``candidates`` != 0; < 0: its own caps); and through the
    fused kernels (K2 with checkpoints + K2b), which is what ``install()`` makes ``SpinArray.applypulse`` run.
    Returns the JSON object of the run (no cpu_baseline)."""
    import mrphy_amd
    from mrphy_amd import beffective, sims, synth, interp, fused, workspace
    dev = torch.device('cuda', torch.cuda.current_device())
    nM = n ** 3
    sp = synth.cube_spins(n, dtype=torch.float32, device=dev)
    if multi:
        p = synth.pulse(nT // 2, dtype=torch.float32, device=dev, dt=8e-6)
        dt_fine = torch.tensor([4e-6], dtype=torch.float32, device=dev)
    else:
        p = synth.pulse(nT, dtype=torch.float32, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    # launch durations of the kernels of the gradient routes, measured where they are launched: HIP events on the
    # launch stream (torch's current stream) immediately around the C-ABI call
    lib = mrphy_amd.require_library()
    launches = {'mrphy_blochsim_fwd_parts': [], 'mrphy_blochsim_bwd_parts': [], 'mrphy_blochsim_rfgr_bwd': []}

    def timed_entry(name):
        fn = getattr(lib, name)

        def call(*args):
            e0, e1 = ev(), ev()
            e0.record(); rc = fn(*args); e1.record()
            launches[name].append((e0, e1))
            return rc
        return fn, call
    saved = {nm: timed_entry(nm) for nm in launches}

    def fine(rf, gr):
        if not multi:
            return rf, gr, p['dt']
        return interp.interpT(rf, gr, p['dt'], dt_fine)

    def mean_ms(name, last):
        evs = launches[name][-last:] if last else []
        return sum(x.elapsed_time(y) for x, y in evs) / len(evs) if evs else None

    ss = nM * nT
    # K1h: reads Beff, writes the history (24 B/spin-step) + Mi, Mo and three constants per spin;
    # K3: reads Beff and the history, writes grad_Beff (36 B/spin-step) + gMo, gMi and three constants per spin
    k1h_bytes = 24 * nM * nT + nM * (12 + 12 + 12)
    k3_bytes = 36 * nM * nT + nM * (12 + 12 + 12)

    def materialised(ws):
        r"""W + K iterations of the materialised route; `ws` = a GradWorkspace or None (the caching allocator)."""
        acc = {'interpT+K0_rfgr2beff': [], 'K1_fwd_history': [], 'backward (K3, K0 adjoint, interpT adjoint)': []}
        tot, g = [], None
        for nm in launches:
            launches[nm].clear()
        for it in range(W + K):
            rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
            e = [ev() for _ in range(4)]
            e[0].record()
            rf_f, gr_f, dt_f = fine(rf, gr)
            assert rf_f.shape[2] == nT, (rf_f.shape, nT)
            beff = beffective.rfgr2beff(rf_f, gr_f, sp['loc'], Δf=sp['Δf'], γ=sp['γ'],
                                        out=None if ws is None else ws.beff)
            e[1].record()
            Mo = sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=dt_f, workspace=ws)
            e[2].record()
            Mo.sum().backward()
            e[3].record()
            torch.cuda.synchronize()
            if it >= W:
                for k_, (i, j) in zip(acc, ((0, 1), (1, 2), (2, 3))):
                    acc[k_].append(e[i].elapsed_time(e[j]))
                tot.append(e[0].elapsed_time(e[3]))
            g = (rf.grad, gr.grad)
            del beff, Mo
        k1h_ms, k3_ms = mean_ms('mrphy_blochsim_fwd_parts', K), mean_ms('mrphy_blochsim_bwd_parts', K)
        # medians over the K iterations: a single iteration that has to wait for the allocator (a 6.4-GB hipMalloc is
        # milliseconds of host time) would otherwise set the mean of its stage
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        return {'spin_steps_per_s_fwd_bwd': ss / (med(tot) * 1e-3), 'ms_total': med(tot), 'ms_total_max': max(tot),
                'stages_ms': {k_: med(v) for k_, v in acc.items()},
                'K1h_launch_ms': k1h_ms, 'K1h_frac_hbm': k1h_bytes / (k1h_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'K3_launch_ms': k3_ms, 'K3_frac_hbm': k3_bytes / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}, g

    for nm, (fn, call) in saved.items():
        setattr(lib, nm, call)
    mat = mat_ws = ws_report = None
    g_mat = g_ws = None
    try:
        if not fused_only:
            if route != 'workspace' or candidates == 0:
                mat, g_mat = materialised(None)
                from mrphy_amd import _hist
                mat['blocks'] = ('history, grad_Beff and Beff from the caching allocator (the reference signature as it '
                                 f"is); the history in {_hist.n_parts_for(mrphy_amd._lib.F32P, 1, nM, nT)} separately "
                                 'allocated parts (mrphy_amd/_hist.py), nothing probed')
            if candidates != 0 and route != 'allocator':
                torch.cuda.empty_cache()
                ws = workspace.GradWorkspace((1, nM, nT, 3), torch.float32, dev,
                                             candidates=None if candidates < 0 else candidates)
                ws_report = ws.report
                if log_:
                    log_(f'grad workspace: {ws.report}')
                mat_ws, g_ws = materialised(ws)
                mat_ws['blocks'] = ('sims.blochsim(..., workspace=ws), rfgr2beff(..., out=ws.beff): the placement-probed '
                                    'blocks of mrphy_amd.workspace.GradWorkspace (an extension of the reference signature)')
                del ws
                torch.cuda.empty_cache()
        f_fwd = f_bwd = 0.
        t_wall = 0.
        launches['mrphy_blochsim_rfgr_bwd'].clear()
        for it in range(W + K):
            if it == W:
                torch.cuda.synchronize()
                t_wall = time.perf_counter()
            rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
            e = [ev() for _ in range(3)]
            e[0].record()
            rf_f, gr_f, dt_f = fine(rf, gr)
            Mo = fused.blochsim_rfgr(sp['M0'], rf_f, gr_f, sp['loc'], Δf=sp['Δf'], γ_beff=sp['γ'],
                                     T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=dt_f)
            e[1].record()
            Mo.sum().backward()
            e[2].record()
            if it >= W:
                launches.setdefault('_fused_iter', []).append((e[0], e[1], e[2]))
            g_fused = (rf.grad, gr.grad)
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - t_wall
        for e0, e1, e2 in launches.pop('_fused_iter'):
            f_fwd += e0.elapsed_time(e1)
            f_bwd += e1.elapsed_time(e2)
    finally:
        for nm, (fn, call) in saved.items():
            setattr(lib, nm, fn)
    rel_l2 = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
    k2b_ms = mean_ms('mrphy_blochsim_rfgr_bwd', K)         # K2b and its second pass
    mode = mrphy_amd.precision.get()
    prof = k2_valu_profile(mode, K2B_PMC)
    k2b = {'kernel': 'k_bloch_rfgr_bwd (+ its second pass): the fused adjoint K2b', 'launch_ms': k2b_ms,
           'bound': 'fp32/fp64 VALU issue', 'spin_steps_per_s': ss / (k2b_ms * 1e-3)}
    if prof is None:
        k2b.update(valu_slot_frac=None, valu_source=rel(K2B_PMC) + ' missing or collected on other kernel '
                                                      'sources (source_id mismatch): not assumed')
    else:
        slots = prof[0] + prof[1]
        k2b.update(valu_insts_per_wave_step=prof[0], half_rate_insts_per_wave_step=prof[1],
                   issue_slots_per_wave_step=slots, valu_slot_frac=slots * ss / (k2b_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                   valu_source=rel(K2B_PMC) + ' (rocprofv3 --pmc SQ_INSTS_VALU + per-type counters over this '
                               'kernel); fraction of the 2.4-GHz lane-op peak')
    best = mat_ws if mat_ws is not None else mat           # the roofline line: the route the bench recommends
    roof = None
    if best is not None:
        k3_ms = best['K3_launch_ms']
        roof = {'kernel': 'k_bloch_bwd_lines (K3: adjoint sweep of the materialised route; reads Beff + history, '
                          'writes grad_Beff)', 'bound': 'hbm', 'achieved': k3_bytes / (k3_ms * 1e-3) / 1e9,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': k3_bytes / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'traffic': None, 'traffic_source': None, 'launch_ms': k3_ms,
                'algorithmic_bytes_per_launch': k3_bytes,
                'blocks': 'GradWorkspace (placement-probed)' if mat_ws is not None else 'caching allocator',
                'K1h': {'kernel': 'k_bloch_fwd_lines<SAVE> (K1h: forward that writes the history)',
                        'launch_ms': best['K1h_launch_ms'], 'frac': best['K1h_frac_hbm'],
                        'algorithmic_bytes_per_launch': k1h_bytes}}
    out = {'metric': 'spin-steps/sec', 'value': ss * K / t_wall, 'unit': 'spin-steps/s',
           'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': 1e3 * t_wall / K, 'higher_is_better': True,
           'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'mode': 'grad',
           'precision': mode,
           'step': 'one pulse-design iteration through the fused kernels (what install() binds to '
                   'SpinArray.applypulse): ' + ('coarse pulse -> interpT -> ' if multi else '')
                   + 'K2 with checkpoints -> sum(Mo).backward() -> K2b -> gradients of the '
                   + ('coarse' if multi else 'fine') + ' rf, gr; wall clock over the K iterations, host time included',
           'config': {'workload': f'{n}^3 spin cube x {nT}-step pulse, fp32'
                      + (f', coarse pulse ({nT // 2} @ 8 us) -> interpT -> ' if multi else ', ')
                      + 'forward + backward to rf/gr', 'baseline_config': baseline_config(n, nT, 'grad', 1),
                      'spins': nM, 'nT': nT},
           'roofline': roof,
           'kernels': {'K2b_fused_adjoint': k2b},
           'materialised': mat, 'materialised_workspace': mat_ws,
           'placement': None if ws_report is None else {'grad_workspace': dict(
               ws_report, what='mrphy_amd.workspace.GradWorkspace: ms of K1h writing its history into the parts '
                               '(K1h_ms[0]) and into each candidate block alone (K1h_ms[1:]), ms of K3 writing grad_Beff '
                               'into each candidate, before the timed iterations; probe_seconds and peak_bytes are what '
                               'the draw cost')},
           'fused': {'ms_fwd_with_checkpoints': f_fwd / K, 'ms_bwd': f_bwd / K,
                     'spin_steps_per_s_fwd_bwd': ss * K / ((f_fwd + f_bwd) * 1e-3),
                     'note': 'K2 (checkpoint every 16 steps) + K2b; VALU-bound, no Beff/history/'
                             'grad_Beff in HBM; deterministic reduction'},
           'grad_fused_vs_materialised_rel_l2': None if (g_mat or g_ws) is None else {
               'rf': rel_l2(g_fused[0], (g_mat or g_ws)[0]), 'gr': rel_l2(g_fused[1], (g_mat or g_ws)[1])},
           'grad_workspace_equals_allocator_bitwise': None if (g_ws is None or g_mat is None) else bool(
               torch.equal(g_ws[0], g_mat[0]) and torch.equal(g_ws[1], g_mat[1]))}
    pj = TRAFFIC
    if out['roofline'] is not None and os.path.exists(pj):
        try:
            w = json.load(open(pj))['workloads'][f'grad_{n}_{nT}']
            k3 = next(v for k_, v in w.items() if k_.startswith('k_bloch_bwd_lines'))
            out['roofline']['traffic'] = k3['total_bytes']
            out['roofline']['traffic_source'] = (rel(TRAFFIC) + ': rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                                                 '(separate runs) over this kernel on this workload -- not collected live')
        except Exception:
            pass
    return out


def grad_mode(a):
    r"""``--mode grad`` / ``--config 4``: :func:`grad_measure` + the CPU leg, one JSON line."""
    protect_stdout()
    from mrphy_amd import synth, fused
    dev = torch.device('cuda', 0)
    n, nT = a.n, a.nT
    out = grad_measure(n, nT, a.steps, a.warmup, multi=not a.no_interp, fused_only=a.fused_only,
                       candidates=a.grad_candidates, log_=log, route=a.grad_route)
    rel_l2 = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
    if not a.no_cpu:
        log(f'cpu baseline (forward + backward) on {host_cores()} cores')
        cb, (c_rf, c_gr), cidx = cpu_baseline_grad(n, nT, min(a.cpu_spins, 8192), a.cpu_chunks, min(a.cpu_budget, 60.0))
        # the same gradient contribution of the same spins through the GPU's fused route, fine pulse
        pf = synth.pulse(nT, dtype=torch.float32, device=dev)
        sps = synth.cube_spins(n, cidx, dtype=torch.float32, device=dev)
        rf, gr = pf['rf'].clone().requires_grad_(True), pf['gr'].clone().requires_grad_(True)
        Mo = fused.blochsim_rfgr(sps['M0'], rf, gr, sps['loc'], Δf=sps['Δf'], γ_beff=sps['γ'], T1=sps['T1'],
                                 T2=sps['T2'], γ=sps['γ'], dt=pf['dt'])
        Mo.sum().backward()
        cb['gpu_vs_cpu_rel_l2_on_sample'] = {'grad_rf': rel_l2(rf.grad.cpu().double(), c_rf.double()),
                                             'grad_gr': rel_l2(gr.grad.cpu().double(), c_gr.double())}
        out['cpu_baseline'] = cb
    emit(out)


def free_port():
    r"""A TCP port that was free a moment ago (bound and released: another process could take it in
    between -- then the rendezvous fails loudly and the launcher returns non-zero; set MASTER_PORT
    to choose one)."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(a):
    r"""``python bench.py --gpus N`` with N > 1 and no rank environment: this process becomes the
    launcher.  BEFORE any GPU call it starts N fresh children of this same script (one process per
    GPU, RANK = LOCAL_RANK = 0..N-1, rendezvous on 127.0.0.1), waits for them, and exits non-zero
    if any child does; rank 0 prints the JSON line on the inherited stdout.  It never re-executes
    itself and never touches the GPU (a process that has initialised the GPU must not exec)."""
    import subprocess
    port = int(os.environ.get('MASTER_PORT', 0)) or free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    log(f'launcher: started {a.gpus} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}')
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            c = p.poll()
            if c is None:
                continue
            pending.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in pending:            # a rank died: end the others (exact pids) ...
                    q.terminate()
                deadline = time.time() + 20.0
                for q in pending:            # ... and do not let a rank hung in RCCL stall us
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
        time.sleep(0.05)
    if rc:
        log(f'launcher: a rank exited with code {rc}')
    sys.exit(rc if 0 <= rc < 256 else 1)


def main():
    a = parse()
    if a.mode == 'grad':
        return grad_mode(a)
    if a.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(a)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.arena is None:
        a.arena = 0 if world == 1 else 3
    if a.dry_launch:
        print(json.dumps({'dry_launch': True, 'rank': rank, 'local_rank': local, 'world_size': world,
                          'gpus': a.gpus, 'master': f"{os.environ.get('MASTER_ADDR')}:"
                                                    f"{os.environ.get('MASTER_PORT')}"}), flush=True)
        return
    if world != a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}')
    protect_stdout()                 # every rank: eight banners must not reach stdout either
    assert torch.cuda.is_available(), 'bench.py needs the GPU (no CPU fallback)'
    # MRPHY_BENCH_REHEARSE=gloo: a CODE-PATH rehearsal of the N-rank run on a box with fewer GPUs than ranks --
    # the ranks share the visible GPUs (rank % device_count) and talk over gloo instead of RCCL (which refuses
    # two ranks on one device).  Everything else is the real path: launcher, rank environment, sharding by
    # rank, the kernels, the asynchronous all-gather, the MAX-reduced clock, the one JSON line.  Its timing
    # is NOT a multi-GPU measurement and the JSON line says so.
    rehearse = os.environ.get('MRPHY_BENCH_REHEARSE', '')
    if rehearse not in ('', 'gloo'):
        sys.exit(f'bench.py: MRPHY_BENCH_REHEARSE={rehearse!r} (only "gloo")')
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # under torch.distributed.run (RANK set) the process group is always brought up, also for one
    # rank: the RCCL init and the all-gather path are then exercised on a single-GPU box too
    use_dist = world > 1 or 'RANK' in os.environ
    if a.shard_of and world != 1:
        sys.exit('bench.py: --shard-of is a one-GPU rehearsal (use it with --gpus 1)')
    if a.shard_of and not use_dist:       # the rehearsal takes the RCCL path: a one-rank group
        os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(int(os.environ.get('MASTER_PORT', 0)) or free_port()))
        use_dist = True
    if use_dist:
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    import mrphy_amd
    from mrphy_amd import beffective, sims, fused, synth, workspace
    from mrphy_amd.dist import shard_bounds, all_gather_spins
    mrphy_amd.require_library()
    log(f'rank {rank}/{world} on {torch.cuda.get_device_name(dev)}')
    ccomm = None
    if use_dist and a.collectives == 'c-abi' and not rehearse:
        from mrphy_amd.dist import CComm, use_c_abi
        ccomm = CComm(world, rank, dev)              # rendezvous over the process group just brought up
        use_c_abi(ccomm)

    n, nT, K, W = a.n, a.nT, a.steps, a.warmup
    nM = n ** 3
    lo, hi = shard_bounds(nM, world, rank)
    p = synth.pulse(nT, dtype=torch.float32, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def run_block(lo, hi, gather_nM, *, n=n, nT=nT, p=p, K=K, W=W, arena_c=a.arena):
        r"""W warm-up + K timed steps of the hot path over spins [lo, hi) of the n^3 cube; with a process
        group, every step's Mo goes through the asynchronous all-gather into a `gather_nM`-spin
        result.  `arena_c` > 0: the step's Beff block is the fastest of that many candidates
        (workspace.BeffArena, rfgr2beff(..., out=, store=)); 0: the plain reference signature, a fresh allocation
        per step.  Returns (seconds for the K steps, K0 ms, K1 ms, last result, the spins' maps, the arena)."""
        idx = torch.arange(lo, hi, device=dev)
        sp = synth.cube_spins(n, idx, dtype=torch.float32, device=dev)
        k0_ev, k1_ev = [], []
        arena = None
        if arena_c > 0:                   # before the timed region: pick the block the step runs fastest on
            def probe(b, store):          # two arguments: the arena times K0's store policies as well
                beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'], out=b, store=store)
                sims.blochsim(sp['M0'], b, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
            arena = workspace.BeffArena((1, hi - lo, nT, 3), torch.float32, dev, probe, candidates=arena_c)
            log(f'arena: {arena.report}')

        def step(timed):
            e = [ev() for _ in range(3)] if timed else None
            if timed:
                e[0].record()
            beff = beffective.rfgr2beff(p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ'],
                                        out=None if arena is None else arena.block,
                                        store=None if arena is None else arena.store)
            if timed:
                e[1].record()
            Mo = sims.blochsim(sp['M0'], beff, T1=sp['T1'], T2=sp['T2'], γ=sp['γ'], dt=p['dt'])
            if timed:
                e[2].record()
                k0_ev.append((e[0], e[1]))
                k1_ev.append((e[1], e[2]))
            # gather BEFORE releasing Beff: otherwise the caching allocator carves the gather's small
            # buffers out of the freed 103 GB block and the next step has to allocate a new one.
            # The collective is asynchronous: it overlaps with the next step's rfgr2beff on the compute
            # stream and is waited for (stream-level) before the step after that, and at the fence.
            out = all_gather_spins(Mo, gather_nM, force=True, async_op=True) if use_dist else Mo
            del beff
            return out

        def fence():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        def finish(x):
            return x.result() if use_dist else x

        with torch.no_grad():
            Mo = None
            for _ in range(W):
                prev, Mo = Mo, step(False)
                if prev is not None:
                    finish(prev)
            if Mo is not None:
                Mo = finish(Mo)
            fence()
            t0 = time.perf_counter()
            pend = None
            for _ in range(K):
                prev, pend = pend, step(True)
                if prev is not None:
                    finish(prev)                 # the previous step's gather, one step later
            Mo = finish(pend)                    # the last gather completes inside the timed region
            fence()
            elapsed = time.perf_counter() - t0
        k0 = sum(s_.elapsed_time(e_) for s_, e_ in k0_ev) / max(len(k0_ev), 1)
        k1 = sum(s_.elapsed_time(e_) for s_, e_ in k1_ev) / max(len(k1_ev), 1)
        return elapsed, k0, k1, Mo, sp, arena

    rows = hi - lo
    shard = None
    if world == 1 and a.shard_of:
        # FIRST, in memory this process has not touched yet -- as in a rank's own fresh process: blocks
        # re-allocated from freed device memory are the slow-to-write kind (tools/archive/block_probe.py: the same
        # virtual address written in 1.86 ms before a free / re-allocate and in 2.13 ms after,
        # profiles/r04_block_probe.json), which a real rank does not see
        slo, shi = shard_bounds(nM, a.shard_of, 0)
        log(f'shard rehearsal: rank 0 of {a.shard_of}: spins [{slo}, {shi})')
        el_s, k0_s, k1_s, Mo_s, _, arena_s = run_block(slo, shi, shi - slo, arena_c=a.arena or 3)   # as a rank of N > 1 would
        assert Mo_s.shape == (1, shi - slo, 3)
        shard = (slo, shi, el_s, k0_s, k1_s, None if arena_s is None else arena_s.report)
        del Mo_s, arena_s
        torch.cuda.empty_cache()
    # The step through the PLAIN reference signatures (no out=, no store=: a fresh Beff per step from the caching allocator),
    # FIRST, in memory the process has not used yet -- what a user's own process gets.  Its block stays
    # in torch's cache and is the first candidate the arena below draws, so the headline loses nothing by the order.
    plain, Mo_plain = None, None
    if world == 1 and a.arena > 0:
        try:
            log('plain-signature step')
            Kp = max(1, min(K, 5))
            el_p, k0_p, k1_p, Mo_plain, _, _ = run_block(lo, hi, nM, K=Kp, W=2, arena_c=0)
            plain = {'ms_per_step': 1e3 * el_p / Kp, 'K0_ms': k0_p, 'K1_ms': k1_p, 'steps': Kp, 'warmup': 2,
                     'what': 'rfgr2beff(rf, gr, loc, Δf=, γ=) -> sims.blochsim(Mi, Beff, T1=, T2=, γ=, dt=): exactly the '
                             'reference signatures, Beff a fresh tensor per step; measured before the arena and the headline '
                             'region, in memory the process had not used'}
        except Exception as e:  # noqa: BLE001
            log(f'plain-signature leg failed: {type(e).__name__}: {e}')
    log('inputs resident; warmup + timed region')
    elapsed, k0_ms, k1_ms, Mo, sp, arena = run_block(lo, hi, nM)
    if plain is not None:
        plain['equals_arena_result_bitwise'] = bool(torch.equal(Mo_plain, Mo))
    del Mo_plain
    per_rank_ms = [1e3 * elapsed / K]
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [1e3 * float(x) / K for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    assert Mo.shape == (1, nM, 3) and bool(torch.isfinite(Mo).all())
    # every rank holds the same gathered result, bit for bit (a checksum of the bit patterns, compared across
    # ranks); with each rank's own slice checked against the fused kernel below, every copy is then right
    gathered_consistent = True
    if use_dist and world > 1:
        chk = Mo.contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        gathered_consistent = all(int(c) == int(allchk[0]) for c in allchk)

    log(f'{K} steps in {elapsed:.3f}s')

    # fused rf,gr -> Mo (K2): same workload, no Beff in HBM; VALU-bound, reported beside
    k2_ms = k2_fast_ms = None
    if not a.no_fused:
        with torch.no_grad():
            f = lambda: fused.blochsim_rfgr(sp['M0'], p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'],  # noqa
                                            γ_beff=sp['γ'], T1=sp['T1'], T2=sp['T2'], γ=sp['γ'],
                                            dt=p['dt'])

            def timed_fused():
                Mf = f()
                torch.cuda.synchronize()
                e0, e1 = ev(), ev()
                e0.record()
                for _ in range(max(K // 2, 1)):
                    Mf = f()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / max(K // 2, 1), Mf
            k2_ms, Mf = timed_fused()
            fused_equal = bool((Mf == (Mo[:, lo:hi] if world > 1 else Mo)).all())
            if use_dist and world > 1:          # every rank's own slice of its gathered copy, AND-ed over ranks
                fe = torch.tensor([1 if fused_equal else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(fe, op=dist.ReduceOp.MIN)
                fused_equal = bool(int(fe))
            if mrphy_amd.precision.get() == 'precise':      # the all-fp32 step beside it
                with mrphy_amd.precision('fast'):
                    k2_fast_ms, _ = timed_fused()

    # What this process's 103-GB block sustains for a plain streaming write (torch fill_ on the very
    # block the steps used -- the caching allocator hands it out again): VRAM regions differ by
    # ~15 % in write rate (and, inversely, ~8 % in read rate) depending on where the driver placed
    # the allocation (docs/LABNOTES.md "Placement"), which is what K0's and K1's
    # process-to-process spread comes from.
    placement = None
    if world == 1:
        with torch.no_grad():
            blk = arena.block if arena is not None else torch.empty((1, nM, nT, 3), dtype=torch.float32, device=dev)
            tw = []
            for _ in range(3):
                e0, e1 = ev(), ev()
                e0.record(); blk.fill_(0.5); e1.record()
                torch.cuda.synchronize()
                tw.append(e0.elapsed_time(e1))
            placement = {'beff_block_fill_TBps': blk.numel() * 4 / min(tw[1:]) / 1e9,
                         'note': 'plain torch streaming write (fill_) of the Beff block of this process: '
                                 '~6.9 on a fast-to-write placement, ~6.0 on a slow one (which in turn '
                                 'reads ~8 % faster: K1 gains what K0 loses)',
                         'arena': None if arena is None else dict(
                             arena.report, what='mrphy_amd.workspace.BeffArena: candidate blocks x K0 store policies, ms of one step '
                                                '(K0 + K1) on each before the timed region, the fastest kept')}
            del blk

    if rank != 0:
        if ccomm is not None:
            ccomm.destroy()
        dist.destroy_process_group()
        return

    # ---- beside the headline, in the same process and the same JSON line (N = 1): the step through the PLAIN
    # reference signature, and the other single-GPU BASELINE configs (VERDICT r4, items 2 and "weak 4/5") ----------
    extra = arena_leg = None
    if world == 1:
        fused_equal_main = fused_equal if k2_ms is not None else None
        Mo_keep, sp_keep = Mo, sp
        del arena
        torch.cuda.empty_cache()
        extras_error = None
        try:          # the headline line must come out whatever happens to the additional legs
            if a.arena == 0 and not a.no_extra_configs:
                log('arena step (the same step with a probed Beff block: rfgr2beff(..., out=, store=))')
                Ka = max(1, min(K, 5))
                el_a, k0_a, k1_a, Mo_a, _, ar = run_block(lo, hi, nM, K=Ka, W=2, arena_c=3)
                arena_leg = {'ms_per_step': 1e3 * el_a / Ka, 'K0_ms': k0_a, 'K1_ms': k1_a, 'steps': Ka, 'warmup': 2,
                             'equals_headline_result_bitwise': bool(torch.equal(Mo_a, Mo_keep)), 'arena': ar.report,
                             'what': 'the step with rfgr2beff(..., out=block, store=policy): the block and K0\'s store policy '
                                     'chosen by mrphy_amd.workspace.BeffArena among candidates (an extension of the reference '
                                     'signature); measured after the headline region, in memory the process has used before'}
                del Mo_a, ar
                torch.cuda.empty_cache()
            if not a.no_extra_configs and (n, nT) == (128, 4096):
                extra = {}
                log('configs[1]: 64^3 x 1024')
                n1, nT1 = 64, 1024
                p1 = synth.pulse(nT1, dtype=torch.float32, device=dev)
                K1n = max(K, 10)
                el1, k0_1, k1_1, Mo1, sp1, ar1 = run_block(0, n1 ** 3, n1 ** 3, n=n1, nT=nT1, p=p1, K=K1n, W=max(W, 3))
                with torch.no_grad():
                    f1 = lambda: fused.blochsim_rfgr(sp1['M0'], p1['rf'], p1['gr'], sp1['loc'], Δf=sp1['Δf'],  # noqa: E731
                                                     γ_beff=sp1['γ'], T1=sp1['T1'], T2=sp1['T2'], γ=sp1['γ'], dt=p1['dt'])
                    Mf1 = f1()
                    torch.cuda.synchronize()
                    e0, e1 = ev(), ev()
                    e0.record()
                    for _ in range(K1n):
                        Mf1 = f1()
                    e1.record()
                    torch.cuda.synchronize()
                    k2_1 = e0.elapsed_time(e1) / K1n
                r1 = n1 ** 3
                b1k1, b1k0 = 12 * r1 * nT1 + r1 * 36, 12 * r1 * nT1 + r1 * 16
                prof1 = k2_valu_profile(mrphy_amd.precision.get())
                extra['1'] = {
                    'workload': f'{n1}^3 spin cube ({r1} spins) x {nT1}-step pulse, fp32: rfgr2beff + sims.blochsim per step',
                    'baseline_config': baseline_config(n1, nT1, 'fwd', 1), 'steps': K1n,
                    'ms_per_step': 1e3 * el1 / K1n, 'value': r1 * nT1 * K1n / el1, 'unit': 'spin-steps/s',
                    'roofline': {'kernel': 'k_bloch_fwd (K1)', 'bound': 'hbm', 'launch_ms': k1_1,
                                 'achieved': b1k1 / (k1_1 * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': b1k1 / (k1_1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': b1k1},
                    'K0_rfgr2beff': {'ms': k0_1, 'frac_hbm': b1k0 / (k0_1 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    'K2_fused_rfgr_fwd': {'ms': k2_1, 'spin_steps_per_s': r1 * nT1 / (k2_1 * 1e-3),
                                          'valu_slot_frac': None if prof1 is None else
                                          (prof1[0] + prof1[1]) * r1 * nT1 / (k2_1 * 1e-3) / VALU_PEAK_LANE_OPS,
                                          'equals_K0_K1_bitwise': bool(torch.equal(Mf1, Mo1))},
                    'arena': None if ar1 is None else ar1.report}
                del Mo1, Mf1, sp1, ar1
                torch.cuda.empty_cache()
                log('configs[4]: 64^3 x 2048, coarse pulse -> interpT -> forward + backward')
                g4 = grad_measure(64, 2048, max(3, min(K, 10)), 2, candidates=a.grad_candidates, log_=log)
                extra['4'] = {
                    'workload': g4['config']['workload'], 'baseline_config': g4['config']['baseline_config'],
                    'steps': g4['steps'], 'ms_per_iter': g4['ms_per_step'], 'value': g4['value'],
                    'unit': 'spin-steps/s (forward + backward, fused kernels, wall clock)',
                    'fused': g4['fused'], 'K2b': {k_: g4['kernels']['K2b_fused_adjoint'].get(k_) for k_ in
                                                  ('launch_ms', 'valu_slot_frac', 'spin_steps_per_s')},
                    'materialised': g4['materialised'], 'materialised_workspace': g4['materialised_workspace'],
                    'roofline': g4['roofline'], 'grad_workspace': (g4['placement'] or {}).get('grad_workspace'),
                    'grad_fused_vs_materialised_rel_l2': g4['grad_fused_vs_materialised_rel_l2'],
                    'grad_workspace_equals_allocator_bitwise': g4['grad_workspace_equals_allocator_bitwise']}
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            extras_error = f'{type(e).__name__}: {e}'
            log(f'additional legs failed: {extras_error}')
            torch.cuda.empty_cache()
        Mo, sp = Mo_keep, sp_keep
        fused_equal = fused_equal_main

    ss_total = nM * nT
    # algorithmic HBM bytes of one K1 launch on this rank: Beff read + Mi, Mo + per-spin E1,E2,E1-1
    k1_bytes = 12 * rows * nT + rows * (12 + 12 + 3 * 4)
    k0_bytes = 12 * rows * nT + rows * (12 + 4)
    out = {
        'metric': 'spin-steps/sec', 'value': ss_total * K / elapsed, 'unit': 'spin-steps/s',
        'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': 1e3 * elapsed / K,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'precision': mrphy_amd.precision.get() + ' (fp32 step; see mrphy_amd/_host.py: precision)',
        'per_rank_ms_per_step': per_rank_ms,
        'rccl_ranks': (world if use_dist else 0) if not rehearse else 0,
        'collectives': None if not use_dist else ('libmrphy_comm.so (C ABI, RCCL direct)' if ccomm is not None
                                                  else 'torch.distributed (' + ('gloo' if rehearse else 'nccl = RCCL') + ')'),
        'gathered_result_identical_on_all_ranks': gathered_consistent,
        'config': {'workload': f'{n}^3 spin cube ({nM} spins) x {nT}-step pulse, fp32: '
                               f'rfgr2beff + sims.blochsim per step'
                               + (f', spins sharded over {world} GPUs + RCCL all-gather of Mo'
                                  if world > 1 else ''),
                   'baseline_config': baseline_config(n, nT, 'fwd', world),
                   'spins': nM, 'nT': nT, 'parallelism': f'spins/{world}'},
        'roofline': {'kernel': 'k_bloch_fwd (K1, blochsim forward over materialised Beff)',
                     'bound': 'hbm', 'achieved': k1_bytes / (k1_ms * 1e-3) / 1e9,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': k1_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'traffic': None, 'traffic_source': None, 'launch_ms': k1_ms,
                     'algorithmic_bytes_per_launch': k1_bytes,
                     'spin_steps_per_s': rows * nT / (k1_ms * 1e-3)},
        'kernels': {
            'K0_rfgr2beff': {'ms': k0_ms, 'GBps': k0_bytes / (k0_ms * 1e-3) / 1e9,
                             'frac_hbm': k0_bytes / (k0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            'K1_blochsim_fwd': {'ms': k1_ms, 'GBps': k1_bytes / (k1_ms * 1e-3) / 1e9},
        },
    }
    if rehearse:
        out['rehearsal'] = (f'MRPHY_BENCH_REHEARSE={rehearse}: {world} ranks sharing {torch.cuda.device_count()} GPU(s) '
                            f'over gloo -- a rehearsal of the N-rank code path; value / ms_per_step are NOT a '
                            f'multi-GPU measurement')
    if placement is not None:
        out['placement'] = placement
    if plain is not None:
        out['plain_signature_ms_per_step'] = plain['ms_per_step']
        out['plain_signature'] = plain
    elif world == 1 and a.arena == 0:
        out['plain_signature_ms_per_step'] = out['ms_per_step']         # the headline IS the plain signature
        out['headline_api'] = ('rfgr2beff(rf, gr, loc, Δf=, γ=) -> sims.blochsim(Mi, Beff, T1=, T2=, γ=, dt=): exactly the '
                               'reference signatures, a fresh Beff per step from the caching allocator')
    if arena_leg is not None:
        out['arena_ms_per_step'] = arena_leg['ms_per_step']
        out['arena_step'] = arena_leg
    if extra:
        out['configs'] = extra
    if world == 1 and extras_error:
        out['configs_error'] = extras_error
    if k2_ms is not None:
        def k2_entry(ms, mode):
            e = {'ms': ms, 'spin_steps_per_s': rows * nT / (ms * 1e-3)}
            prof = k2_valu_profile(mode)
            if prof is None:
                e.update(valu_insts_per_wave_step=None, valu_slot_frac=None,
                         valu_source=rel(K2_PMC) + ' missing or collected on other kernel sources '
                                     '(source_id mismatch): not assumed')
                return e
            insts, half = prof
            # VALU-slot fraction = issue slots per wave-step x 64 lanes x wave-steps/s over
            # the fp32 lane-op peak (256 CU x 128 lanes x 2.4 GHz = 78.6e12); an fp64 FMA or an
            # fp32<->fp64 conversion issues at half rate, i.e. takes two slots
            slots = insts + half
            e.update(valu_insts_per_wave_step=insts, half_rate_insts_per_wave_step=half,
                     issue_slots_per_wave_step=slots,
                     valu_slot_frac=slots * rows * nT / (ms * 1e-3) / VALU_PEAK_LANE_OPS,
                     valu_source=rel(K2_PMC) + ' (rocprofv3 --pmc SQ_INSTS_VALU + per-type '
                                 'counters over this kernel); fraction of the 2.4 GHz peak -- the part '
                                 'sustains ~2.1 GHz under this load')
            return e
        mode = mrphy_amd.precision.get()
        K2 = k2_entry(k2_ms, mode)
        K2.update(equals_K0_K1_bitwise=fused_equal, precision=mode,
                  note='VALU-bound (no Beff in HBM); effective 12 B/ss-equivalent bandwidth '
                       f'{12 * rows * nT / (k2_ms * 1e-3) / 1e9:.0f} GB/s is NOT HBM traffic')
        if mode == 'precise':
            K2['fast_step'] = k2_entry(k2_fast_ms, 'fast')
            K2['fast_step']['note'] = ("MRPHY_PRECISION=fast / mrphy_amd.precision('fast'): the all-fp32 "
                                       'step, 2.4e-5 from exact arithmetic on this workload (precise: 1.7e-6)')
        out['kernels']['K2_fused_rfgr_fwd'] = K2
    # HBM bytes per K1 launch from the rocprofv3 PMC passes (profiles/rNN_traffic.json, written by
    # tools/collect_rNN.py): valid for the workload they were collected on only
    tj = TRAFFIC
    wl = {(128, 4096): 'fwd_128_4096', (64, 1024): 'fwd_64_1024'}.get((n, nT)) if world == 1 else None
    if os.path.exists(tj) and wl:
        try:
            w = json.load(open(tj))['workloads'][wl]
            k1 = next(v for k_, v in w.items() if k_.startswith('k_bloch_fwd_lines'))
            out['roofline']['traffic'] = k1['total_bytes']
            out['roofline']['traffic_source'] = (
                rel(TRAFFIC) + ': rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs; '
                'KiB units; FETCH_SIZE doubled per the guide\'s gfx950 rule) over this kernel on this '
                'workload -- not collected live in this run')
        except Exception:
            pass
    # ... and the same kernel's launch durations in the committed rocprofv3 kernel trace of this command, grouped by
    # grid size (profiles/rNN_bench_cfg2_by_size.json, tools/collect_profiles.py): the live HIP-event figure above and the
    # profiler's mean must agree (the per-name `--stats` row averaged three problem sizes)
    bs = newest_profile('bench_cfg2_by_size.json')
    if os.path.exists(bs) and world == 1 and (n, nT) == (128, 4096):
        try:
            grp = next(e for e in json.load(open(bs))['groups'] if e.get('family') == 'K1' and e['grid_work_items'] == rows)
            out['roofline']['rocprof'] = {
                'source': rel(bs) + ': rocprofv3 --kernel-trace of `bench.py --no-extra-configs`, launches of this kernel at '
                          'this grid size only -- a committed collection, not this run',
                'calls': grp['calls'], 'mean_ms': grp['mean_ms'], 'median_ms': grp['median_ms'], 'min_ms': grp['min_ms'],
                'frac_of_mean': grp['frac_hbm_of_mean'], 'frac_of_median': grp['frac_hbm_of_median'],
                'launch_ms_live_over_rocprof_mean': k1_ms / grp['mean_ms']}
        except Exception:
            pass
    log('fused leg done' if k2_ms is not None else 'fused leg skipped')
    if world == 1 and not a.no_cpu:
        log(f'cpu baseline on {host_cores()} cores')
        cb, Mo_cpu, cidx = cpu_baseline(n, nT, a.cpu_spins, a.cpu_chunks, a.cpu_budget)
        d = (Mo[0, cidx.to(dev)].double().cpu() - Mo_cpu[0].double())
        cb['gpu_vs_cpu_rel_l2_on_sample'] = float(d.norm() / Mo_cpu[0].double().norm())
        # ... and both against exact (fp64) integration of the same fp32 field on the same sample
        # (oracle/bloch_c.c), each with the fp32 constants its own run used (exp() on the device / on
        # the host: they differ by an ulp on ~13 % of the spins, which over 4096 steps is more than
        # the arithmetic error of either run).  The GPU-vs-CPU distance is the CPU path's own noise.
        import bloch_c as C
        spc, pc = synth.cube_spins(n, cidx, dtype=torch.float32), synth.pulse(nT, dtype=torch.float32)
        spd = {k: v.to(dev) for k, v in spc.items()}

        def exact_with(consts_dev):
            g_, E1_, E2_, E1m1_ = consts_dev
            return C.blochsim_rfgr(spc['M0'], pc['rf'], pc['gr'], spc['loc'], Δf=spc['Δf'], γ_beff=spc['γ'],
                                   consts=C.constants_from(g_, E1_, E2_, E1m1_, N=1, nM=cidx.numel()),
                                   field_f32=True)[0]
        rl = lambda x, ex: float((x.double().cpu() - ex).norm() / ex.norm())  # noqa: E731
        ex_gpu = exact_with(sims.relax_constants(spd['T1'], spd['T2'], spd['γ'], p['dt'], 4, dev))
        ex_cpu = exact_with(sims.relax_constants(spc['T1'], spc['T2'], spc['γ'], pc['dt'], 4, torch.device('cpu')))
        cb['gpu_vs_exact_rel_l2_on_sample'] = rl(Mo[0, cidx.to(dev)], ex_gpu)
        cb['cpu_vs_exact_rel_l2_on_sample'] = rl(Mo_cpu[0], ex_cpu)
        cb['exact'] = ('oracle/bloch_c.c: fp64 integration of the same fp32 field with the fp32 constants of '
                       'the run it is compared with')
        out['cpu_baseline'] = cb
    if world == 1 and a.verify_full:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import bloch_c as C
        log('whole-problem check against oracle/bloch_c.c')
        spc, pc = synth.cube_spins(n, dtype=torch.float32), synth.pulse(nT, dtype=torch.float32)
        with mrphy_amd.constants_on('cpu'):
            g, E1, E2, E1_1 = sims.relax_constants(spc['T1'], spc['T2'], spc['γ'], pc['dt'], 4, dev)
            Mo_c = sims.blochsim_consts(sp['M0'], beffective.rfgr2beff(
                p['rf'], p['gr'], sp['loc'], Δf=sp['Δf'], γ=sp['γ']),
                γ2πdt=g, E1=E1, E1_1=E1_1, E2=E2)
        torch.set_num_threads(host_cores())
        t0 = time.perf_counter()
        want = C.blochsim_rfgr(spc['M0'], pc['rf'], pc['gr'], spc['loc'], Δf=spc['Δf'], γ_beff=spc['γ'],
                               consts=C.constants_from(g, E1, E2, E1_1, N=1, nM=nM), field_f32=True)
        d = Mo_c.double().cpu() - want
        out['verify_full'] = {'rel_l2': float(d.norm() / want.norm()), 'max_abs': float(d.abs().max()),
                              'spins': nM, 'nT': nT, 'oracle': 'oracle/bloch_c.c, fp64 integration of the '
                              'same fp32 field, same fp32 constants',
                              'oracle_seconds': round(time.perf_counter() - t0, 1)}
    if shard is not None:
        S = a.shard_of
        slo, shi, el_s, k0_s, k1_s, arena_rep = shard
        ms_full, ms_shard = 1e3 * elapsed / K, 1e3 * el_s / K
        out['shard_rehearsal'] = {
            'shard_of': S, 'spins': shi - slo, 'tiles': (shi - slo + 63) // 64,
            'ms_per_step_full': ms_full, 'ms_per_step_shard': ms_shard,
            'K0_ms_shard': k0_s, 'K1_ms_shard': k1_s,
            'K0_frac_hbm_shard': (12 * (shi - slo) * nT + (shi - slo) * 16) / (k0_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'K1_frac_hbm_shard': (12 * (shi - slo) * nT + (shi - slo) * 36) / (k1_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'expected_speedup': ms_full / ms_shard,
            'arena': arena_rep,
            'note': f'ONE rank of an {S}-GPU run rehearsed on one GPU, before the full run (RCCL at world size 1; the gather '
                    f'moves this rank\'s {(shi - slo) * 12 / 1e6:.1f} MB, not the {nM * 12 / 1e6:.1f} MB a real run '
                    'receives): an estimate of the scaling if every rank matches it, NOT a measurement'}
    emit(out)
    if ccomm is not None:
        ccomm.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
