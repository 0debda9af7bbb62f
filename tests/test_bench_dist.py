r"""The rank path of ``bench.py`` and ``mrphy_amd.dist`` on the one GPU a box has: RCCL at world size 1, the one-JSON-line contract, the 3-rank gloo rehearsal.

Regrouped by component in round 5 from ``test_hip_parity.py`` / ``test_hip_round{2,3,4}.py`` (no assertion changed; each test keeps its name).
"""
import pytest

from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------
# multi-GPU path on the real backend (RCCL), one rank
# ---------------------------------------------------------------------------------------------
@pytest.mark.usefixtures('host_constants')
def test_nccl_world1_shard_gather_allreduce_with_hip_kernels():
    r"""`mrphy_amd.dist` on backend `nccl` (= RCCL) on cuda:0 at world_size 1, the shard simulated
    by the HIP kernels (not the oracle): all_gather_spins (forced through the collective, sync and
    async), all_reduce_pulse_grads.  Runs in a child process so that the process group's lifetime
    is its own."""
    import subprocess
    code = r'''
import os, sys
sys.path[:0] = [%r, %r + "/oracle", %r + "/tests"]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
import mrphy_amd
from mrphy_amd import beffective, sims, synth
from mrphy_amd.dist import shard_bounds, all_gather_spins, all_reduce_pulse_grads
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
n, nT = 12, 96
nM = n ** 3
lo, hi = shard_bounds(nM, 1, 0)
sp = synth.cube_spins(n, torch.arange(lo, hi, device=dev), dtype=torch.float32, device=dev, seed_M0=2)
p = synth.pulse(nT, dtype=torch.float32, device=dev)
rf, gr = p["rf"].clone().requires_grad_(True), p["gr"].clone().requires_grad_(True)
beff = beffective.rfgr2beff(rf, gr, sp["loc"], Δf=sp["Δf"], γ=sp["γ"])
Mo = sims.blochsim(sp["M0"], beff, T1=sp["T1"], T2=sp["T2"], γ=sp["γ"], dt=p["dt"])
Mo.sum().backward()
g = all_gather_spins(Mo.detach(), nM, force=True)
h = all_gather_spins(Mo.detach(), nM, force=True, async_op=True).result()
torch.cuda.synchronize()
assert g.shape == (1, nM, 3) and torch.equal(g, Mo.detach()) and torch.equal(h, g)
g_rf, g_gr = rf.grad.clone(), gr.grad.clone()
flat = torch.cat([rf.grad.reshape(-1), gr.grad.reshape(-1)])
dist.all_reduce(flat)                       # the collective itself, on RCCL
all_reduce_pulse_grads(rf.grad, gr.grad)
torch.cuda.synchronize()
assert torch.equal(rf.grad, g_rf) and torch.equal(gr.grad, g_gr)
assert torch.equal(flat[:g_rf.numel()].view_as(g_rf), g_rf)
# against the oracle (CPU)
import bloch_oracle as O
spc = synth.cube_spins(n, dtype=torch.float32, seed_M0=2); pc = synth.pulse(nT, dtype=torch.float32)
with mrphy_amd.constants_on("cpu"):
    Mh = sims.blochsim(sp["M0"], beff.detach(), T1=sp["T1"], T2=sp["T2"], γ=sp["γ"], dt=p["dt"])
ref = O.blochsim(spc["M0"], O.rfgr2beff(pc["rf"], pc["gr"], spc["loc"], Δf=spc["Δf"], γ=spc["γ"]),
                 T1=spc["T1"], T2=spc["T2"], γ=spc["γ"], dt=pc["dt"])
err = float((Mh.cpu().double() - ref.double()).norm() / ref.double().norm())
assert err <= 1e-5, err
dist.destroy_process_group()
print("nccl-ok", err)
''' % (ROOT, ROOT, ROOT, 29500 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'nccl-ok' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_c_abi_collectives_equal_torch_distributed():
    r"""The RCCL entry points behind the C ABI (``include/mrphy_comm.h``: ``mrphy_comm_unique_id / _init /
    _allgather_spins / _allreduce_pulse_grads / _destroy``; VERDICT r5 "missing" #1) at the one rank count a single-GPU
    box allows: (i) a consumer with NO torch.distributed -- plain ctypes on raw device pointers -- gathers and reduces;
    (ii) ``mrphy_amd.dist`` routed through them (``use_c_abi``) returns what its ``torch.distributed`` (backend nccl = RCCL)
    route returns, bit for bit, for the forced world-size-1 collectives, fp32 and fp64, sync and "async" forms.  N > 1
    stays covered by the gloo tests on CPU (``tests/test_dist_gloo.py``): RCCL refuses two ranks on one device."""
    import subprocess
    code = r'''
import ctypes, os, sys
sys.path[:0] = [%r]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import mrphy_amd
from mrphy_amd import _lib, dist as D
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
# (i) ctypes only
lib = _lib.require_comm_library()
idb = ctypes.create_string_buffer(128)
assert lib.mrphy_comm_unique_id(idb) == 0 and any(idb.raw)
h = ctypes.c_void_p()
assert lib.mrphy_comm_init(idb, 1, 0, ctypes.byref(h)) == 0 and h.value
g = torch.Generator().manual_seed(3)
Mo = torch.rand((1, 1000, 3), generator=g).to(dev)
recv = torch.empty_like(Mo)
st = torch.cuda.current_stream().cuda_stream
assert lib.mrphy_comm_allgather_spins(h, Mo.data_ptr(), recv.data_ptr(), Mo.numel(), 0, st) == 0
buf = torch.rand(5 * 256, generator=g, dtype=torch.float64).to(dev); want = buf.clone()
assert lib.mrphy_comm_allreduce_pulse_grads(h, buf.data_ptr(), buf.numel(), 1, st) == 0
assert lib.mrphy_comm_allgather_spins(h, Mo.data_ptr(), recv.data_ptr(), Mo.numel(), 7, st) == -1     # unknown dtype
torch.cuda.synchronize()
assert torch.equal(recv, Mo) and torch.equal(buf, want)
assert lib.mrphy_comm_destroy(h) == 0
# (ii) mrphy_amd.dist through the C ABI == through torch.distributed
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for dt in (torch.float32, torch.float64):
    Mo = torch.rand((2, 777, 3), generator=g, dtype=dt).to(dev)
    grf, ggr = torch.rand((2, 2, 96), generator=g, dtype=dt).to(dev), torch.rand((2, 3, 96), generator=g, dtype=dt).to(dev)
    a = D.all_gather_spins(Mo, 777, force=True)
    a2 = D.all_gather_spins(Mo, 777, force=True, async_op=True).result()
    r1, r2 = grf.clone(), ggr.clone()
    D.all_reduce_pulse_grads(r1, r2, force=True)
    comm = D.CComm(1, 0, dev)
    assert D.use_c_abi(comm) is None
    try:
        b = D.all_gather_spins(Mo, 777, force=True)
        b2 = D.all_gather_spins(Mo, 777, force=True, async_op=True).result()
        s1, s2 = grf.clone(), ggr.clone()
        D.all_reduce_pulse_grads(s1, s2, force=True)
    finally:
        assert D.use_c_abi(None) is comm
        comm.destroy()
    torch.cuda.synchronize()
    assert a.shape == b.shape == (2, 777, 3)
    assert torch.equal(a, b) and torch.equal(a2, b2) and torch.equal(a, Mo)
    assert torch.equal(r1, s1) and torch.equal(r2, s2) and torch.equal(s1, grf)
dist.destroy_process_group()
print("comm-ok")
''' % (ROOT, 29600 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'comm-ok' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.usefixtures('host_constants')
@pytest.mark.parametrize('coll', ['torch', 'c-abi'])
def test_bench_rank_path_prints_exactly_one_json_line(coll):
    r"""bench.py as ONE RANK of a distributed run (RANK / WORLD_SIZE set, as torch.distributed.run and
    bench.py's own launcher set them): RCCL is initialised, the all-gather of Mo and the timing
    exchange run through it -- and stdout carries exactly one line, the JSON, although RCCL writes its
    version banner to stdout at that point (it must arrive on stderr instead).  This is the code path
    of the N > 1 scaling runs, at the one world size a single-GPU box allows."""
    import json
    import subprocess
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29700 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--cube', '16', '--nT', '64',
                        '--steps', '2', '--warmup', '1', '--no-cpu', '--collectives', coll], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, f'stdout must be the JSON line alone, got {len(lines)} lines: {r.stdout[:400]!r}'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and len(d['per_rank_ms_per_step']) == 1
    assert d['collectives'].startswith('libmrphy_comm.so' if coll == 'c-abi' else 'torch.distributed (nccl')
    assert d['gathered_result_identical_on_all_ranks'] is True
    assert d['metric'] == 'spin-steps/sec' and d['value'] > 0 and d['scaling'] == 'strong'
    assert d['kernels']['K2_fused_rfgr_fwd']['equals_K0_K1_bitwise'] is True
    assert 'RCCL version' not in r.stdout
    print('RCCL banner on stderr:', 'RCCL version' in r.stderr)


def test_bench_three_ranks_rehearsed_on_one_gpu():
    r"""The N > 1 code path of ``bench.py`` on a box with one GPU: ``MRPHY_BENCH_REHEARSE=gloo python bench.py --gpus 3``
    -- the launcher starts three rank processes (before any GPU call), every rank simulates its block of the
    spin axis with the HIP kernels (a ragged split: 17^3 = 4913 spins over 3 ranks), the blocks are all-gathered
    (asynchronously, over gloo: RCCL refuses two ranks on one device), the clock is MAX-reduced, rank 0 prints the
    ONE JSON line.  Checked in the line: every rank's gathered copy is the same bit for bit, and each rank's own
    slice of it equals the fused kernel's result bit for bit.  (Three processes on the card: within the box's
    limit of six.)"""
    import json
    import subprocess
    env = dict(os.environ, MRPHY_BENCH_REHEARSE='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--cube', '17', '--nT', '64',
                        '--steps', '2', '--warmup', '1', '--no-cpu'], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, f'stdout must be the JSON line alone, got {len(lines)} lines: {r.stdout[:400]!r}'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 3 and len(d['per_rank_ms_per_step']) == 3 and d['rccl_ranks'] == 0
    assert 'rehearsal' in d and 'NOT a multi-GPU measurement' in d['rehearsal']
    assert d['gathered_result_identical_on_all_ranks'] is True
    assert d['kernels']['K2_fused_rfgr_fwd']['equals_K0_K1_bitwise'] is True
    assert d['config']['spins'] == 17 ** 3 and d['config']['parallelism'] == 'spins/3'
