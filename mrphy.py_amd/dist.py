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


def all_gather_spins(Mo_local: Tensor, nM: int, group=None, force: bool = False) -> Tensor:
    r"""All-gather the per-rank ``(N, nM_r, 3)`` results into ``(N, nM, 3)`` on every rank.

    Blocks may differ by one spin, so shards are padded to the largest block for a single
    ``all_gather_into_tensor`` (one collective, one launch) and trimmed afterwards.
    """
    ws = dist.get_world_size(group)
    if ws == 1 and not force:          # force: run the collective anyway (single-rank rehearsal)
        return Mo_local
    N = Mo_local.shape[0]
    sizes = [shard_bounds(nM, ws, r) for r in range(ws)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = Mo_local.new_zeros((N, mx, 3))
    pad[:, :Mo_local.shape[1]] = Mo_local
    out = Mo_local.new_empty((ws * N, mx, 3))      # concatenated form: accepted by RCCL and gloo
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(ws, N, mx, 3)
    return torch.cat([out[r, :, :hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=1)


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
