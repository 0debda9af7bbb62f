r"""Spin sharding over the GPUs of one node (one process per GPU, ``torch.distributed``).

Spins are independent (time is sequential only within a spin), so the compact spin axis is
cut into ``world_size`` contiguous blocks; every rank simulates its block with the
single-GPU kernels and no data-path collective runs during the time loop.  The only exchange
is ONE all-gather of the final magnetisation (RCCL over xGMI with backend ``nccl``; 128^3
spins: 3.1 MB per rank) -- and, for pulse-design gradients, one all-reduce(sum) of
``grad_rf``/``grad_gr`` (20*nT bytes).  The reference has no distributed code at all
(SURVEY.md §5); this is the MI355X-native scale-out of its spin axis.
"""
from typing import Tuple

import torch
import torch.distributed as dist
from torch import Tensor

__all__ = ['shard_bounds', 'shard_spins', 'all_gather_spins', 'all_reduce_pulse_grads']


def shard_bounds(nM: int, world_size: int, rank: int) -> Tuple[int, int]:
    r"""[lo, hi) of rank's contiguous block; blocks differ by at most one spin."""
    base, rem = divmod(nM, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_spins(x: Tensor, world_size: int, rank: int, dim: int = 1) -> Tensor:
    r"""This rank's block of a per-spin tensor ``(N, nM, ...)`` (a view; broadcast dims of
    size 1 are returned unchanged)."""
    if x.ndim <= dim or x.shape[dim] == 1:
        return x
    lo, hi = shard_bounds(x.shape[dim], world_size, rank)
    return x.narrow(dim, lo, hi - lo)


class _Pending:
    r"""An all-gather in flight: ``.result()`` waits for it (stream-level) and returns `(N, nM, 3)`."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def result(self) -> Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish()


def all_gather_spins(Mo_local: Tensor, nM: int, group=None, force: bool = False,
                     async_op: bool = False):
    r"""All-gather the per-rank ``(N, nM_r, 3)`` results into ``(N, nM, 3)`` on every rank.

    ONE ``all_gather_into_tensor``.  Equal blocks (``nM`` divisible by the world size) are gathered
    straight from ``Mo_local`` and, for ``N == 1``, returned as a view of the receive buffer (no
    pad, no concatenation); blocks that differ by one spin are padded to the largest and trimmed.

    ``async_op=True`` returns a handle whose ``.result()`` waits and yields the tensor, so the
    collective can overlap with the next step's kernels on the compute stream.
    """
    ws = dist.get_world_size(group)
    if ws == 1 and not force:          # force: run the collective anyway (single-rank rehearsal)
        return _Pending(None, lambda: Mo_local) if async_op else Mo_local
    N = Mo_local.shape[0]
    sizes = [shard_bounds(nM, ws, r) for r in range(ws)]
    mx = max(hi - lo for lo, hi in sizes)
    equal = all(hi - lo == mx for lo, hi in sizes)
    if equal:
        send = Mo_local.contiguous()
    else:
        send = Mo_local.new_zeros((N, mx, 3))
        send[:, :Mo_local.shape[1]] = Mo_local
    out = Mo_local.new_empty((ws * N, mx, 3))      # concatenated form: accepted by RCCL and gloo
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)

    def finish():
        o = out.view(ws, N, mx, 3)
        if equal:
            return o.transpose(0, 1).reshape(N, ws * mx, 3)          # a view when N == 1
        return torch.cat([o[r, :, :hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=1)
    if async_op:
        return _Pending(work, finish)
    return finish()


def all_reduce_pulse_grads(*grads: Tensor, group=None):
    r"""Sum ``grad_rf``/``grad_gr`` (replicated pulse, sharded spins) over ranks, in place,
    as ONE flattened all-reduce."""
    ws = dist.get_world_size(group)
    gs = [g for g in grads if g is not None]
    if ws == 1 or not gs:
        return grads
    flat = torch.cat([g.reshape(-1) for g in gs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for g in gs:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
    return grads
