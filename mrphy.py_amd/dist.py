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

__all__ = ['shard_bounds', 'shard_spins', 'all_gather_spins', 'all_reduce_pulse_grads', 'CComm', 'use_c_abi']


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


# ---------------------------------------------------------------------------------------------
# The same two collectives through the C ABI (include/mrphy_comm.h, libmrphy_comm.so): RCCL called directly, on
# the caller's stream -- what a consumer without torch.distributed binds (SURVEY §8b lists these exports).  Here it
# is an opt-in second route, used to check the entry points against torch.distributed bit for bit:
#     comm = CComm(world_size, rank, device)     # rendezvous: the 128-byte id travels over the default process
#     use_c_abi(comm)                            # group (any backend), or is passed in as `unique_id=`
#     all_gather_spins(...), all_reduce_pulse_grads(...)   # now go through mrphy_comm_*
# ---------------------------------------------------------------------------------------------
_C_COMM = None


class CComm:
    r"""A communicator of ``libmrphy_comm.so``: ``mrphy_comm_init`` on ``device`` as ``rank`` of ``nranks``.

    ``unique_id`` (128 bytes from :meth:`new_unique_id` on rank 0, handed over by any channel) -- or ``None``: rank 0
    creates it and broadcasts it over ``torch.distributed``'s ``group`` (gloo works: the id is host bytes)."""

    def __init__(self, nranks: int, rank: int, device, *, unique_id: bytes = None, group=None):
        import ctypes
        from . import _lib
        self.lib = _lib.require_comm_library()
        self.nranks, self.rank, self.device = int(nranks), int(rank), torch.device(device)
        if unique_id is None:
            if nranks == 1:
                unique_id = self.new_unique_id()
            else:
                box = [self.new_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                unique_id = box[0]
        assert len(unique_id) == 128
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check_comm(self.lib.mrphy_comm_init(ctypes.c_char_p(bytes(unique_id)), self.nranks, self.rank,
                                                     ctypes.byref(self._h)), 'mrphy_comm_init')

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check_comm(_lib.require_comm_library().mrphy_comm_unique_id(buf), 'mrphy_comm_unique_id')
        return buf.raw

    @staticmethod
    def _code(x: Tensor) -> int:
        if x.dtype == torch.float32:
            return 0
        if x.dtype == torch.float64:
            return 1
        raise NotImplementedError(f"mrphy_comm: {x.dtype}; float32 and float64 are implemented")

    def all_gather(self, out: Tensor, send: Tensor):
        r"""``out`` (nranks * send.numel() elements, contiguous) <- every rank's ``send``, on torch's current stream."""
        from . import _lib
        assert out.is_contiguous() and send.is_contiguous() and out.numel() == self.nranks * send.numel()
        assert out.device == send.device == self.device or self.device.index is None
        with torch.cuda.device(send.device):
            _lib.check_comm(self.lib.mrphy_comm_allgather_spins(
                self._h, send.data_ptr(), out.data_ptr(), send.numel(), self._code(send),
                torch.cuda.current_stream(send.device).cuda_stream), 'mrphy_comm_allgather_spins')

    def all_reduce_(self, flat: Tensor):
        from . import _lib
        assert flat.is_contiguous()
        with torch.cuda.device(flat.device):
            _lib.check_comm(self.lib.mrphy_comm_allreduce_pulse_grads(
                self._h, flat.data_ptr(), flat.numel(), self._code(flat),
                torch.cuda.current_stream(flat.device).cuda_stream), 'mrphy_comm_allreduce_pulse_grads')

    def destroy(self):
        from . import _lib
        if self._h:
            h, self._h = self._h, None
            with torch.cuda.device(self.device):
                _lib.check_comm(self.lib.mrphy_comm_destroy(h), 'mrphy_comm_destroy')


def use_c_abi(comm: 'CComm' = None):
    r"""Route :func:`all_gather_spins` / :func:`all_reduce_pulse_grads` of this process through ``comm`` (a
    :class:`CComm`); ``None`` restores ``torch.distributed``.  Returns the previous setting."""
    global _C_COMM
    prev, _C_COMM = _C_COMM, comm
    return prev


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
    ws = _C_COMM.nranks if _C_COMM is not None else dist.get_world_size(group)
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
    if _C_COMM is not None:             # the C ABI: RCCL directly, stream-ordered (asynchronous as it is)
        _C_COMM.all_gather(out, send)
        work = None
    else:
        work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)

    def finish():
        o = out.view(ws, N, mx, 3)
        if equal:
            return o.transpose(0, 1).reshape(N, ws * mx, 3)          # a view when N == 1
        return torch.cat([o[r, :, :hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=1)
    if async_op:
        return _Pending(work, finish)
    return finish()


def all_reduce_pulse_grads(*grads: Tensor, group=None, force: bool = False):
    r"""Sum ``grad_rf``/``grad_gr`` (replicated pulse, sharded spins) over ranks, in place,
    as ONE flattened all-reduce (``force``: run the collective at world size 1 too -- a single-rank rehearsal)."""
    ws = _C_COMM.nranks if _C_COMM is not None else dist.get_world_size(group)
    gs = [g for g in grads if g is not None]
    if (ws == 1 and not force) or not gs:
        return grads
    flat = torch.cat([g.reshape(-1) for g in gs])
    if _C_COMM is not None:
        _C_COMM.all_reduce_(flat)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for g in gs:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
    return grads
