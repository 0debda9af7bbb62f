r"""The N > 1 path on CPU: two processes, `gloo`, world_size 2.  The kernels need a GPU, so the
per-rank "simulation" here is the CPU oracle; what is under test is this package's sharding,
all-gather and gradient all-reduce (mrphy_amd/dist.py), i.e. everything bench.py adds for N > 1."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nM, q):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bloch_oracle as O
        from mrphy_amd import synth
        from mrphy_amd.dist import (shard_bounds, shard_spins, all_gather_spins,
                                    all_reduce_pulse_grads)
        torch.set_num_threads(1)
        n, nT = 6, 24
        assert nM <= n ** 3
        full = synth.cube_spins(n, torch.arange(nM), dtype=torch.float64, seed_M0=5)
        p = synth.pulse(nT, dtype=torch.float64)
        lo, hi = shard_bounds(nM, world, rank)
        mine = {k: shard_spins(v, world, rank) for k, v in full.items()}
        assert mine['M0'].shape[1] == hi - lo and mine['γ'].shape == (1, 1)
        rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
        beff = O.rfgr2beff(rf, gr, mine['loc'], Δf=mine['Δf'], γ=mine['γ'])
        Mo = O.blochsim(mine['M0'], beff, T1=mine['T1'], T2=mine['T2'], γ=mine['γ'], dt=p['dt'])
        Mo.sum().backward()
        gathered = all_gather_spins(Mo.detach(), nM)
        pending = all_gather_spins(Mo.detach(), nM, async_op=True)       # the overlapped form
        assert torch.equal(pending.result(), gathered)
        all_reduce_pulse_grads(rf.grad, gr.grad)
        # by value (numpy), not as shared-memory handles that die with this process
        q.put((rank, gathered.numpy().copy(), rf.grad.numpy().copy(), gr.grad.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        return s_.getsockname()[1]


@pytest.mark.parametrize('nM', [64, 101])          # even split, and blocks that differ by one spin
def test_sharded_equals_single_process(nM):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bloch_oracle as O
    from mrphy_amd import synth
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nM, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single-process reference
    n, nT = 6, 24
    full = synth.cube_spins(n, torch.arange(nM), dtype=torch.float64, seed_M0=5)
    p = synth.pulse(nT, dtype=torch.float64)
    rf, gr = p['rf'].clone().requires_grad_(True), p['gr'].clone().requires_grad_(True)
    beff = O.rfgr2beff(rf, gr, full['loc'], Δf=full['Δf'], γ=full['γ'])
    Mo = O.blochsim(full['M0'], beff, T1=full['T1'], T2=full['T2'], γ=full['γ'], dt=p['dt'])
    Mo.sum().backward()
    for rank, gathered, g_rf, g_gr in got:
        gathered, g_rf, g_gr = (torch.from_numpy(x) for x in (gathered, g_rf, g_gr))
        assert gathered.shape == (1, nM, 3)
        assert torch.equal(gathered, Mo.detach()), f'rank {rank}: gathered Mo differs'
        assert torch.allclose(g_rf, rf.grad, rtol=0, atol=1e-12)
        assert torch.allclose(g_gr, gr.grad, rtol=0, atol=1e-12)
