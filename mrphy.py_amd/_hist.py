r"""The history of ``sims.blochsim`` -- the magnetisation before every step, which the adjoint sweep reads back
(reference ``mrphy/sims.py:84-88`` keeps 40 B per spin-step; here 12) -- as one or several separately allocated parts.

The history is internal to the library (its layout is the kernels' own: per 64-spin tile, structure of arrays), so
nothing forces it to be ONE allocation.  Since ABI 5 (``mrphy_blochsim_fwd_parts`` / ``_bwd_parts``) the tiles may be
dealt to up to eight buffers; results are bit-identical however the history is cut.

Why (``profiles/r06_hist_parts_*.json``): how fast a kernel can stream writes into an allocation is a
property of the allocation -- K1h at 64^3 x 2048 runs at 2.45-2.66 ms (0.61-0.66 of HBM peak) when its history is one
block of the slow kind and at 2.08-2.23 ms (0.72-0.77) on one of the fast kind, and which kind a fresh process's
allocator hands out is the box's business (fast for 12 of 44 fresh processes on six boxes).  One allocation is
homogeneous (every pair of places inside a 96-GiB allocation is as slow as the allocation, ``r06_placement_pairs``);
the kind belongs to the allocation, and a history dealt to several allocations is written at the fast rate when they
are not all of one kind: with four parts 26 of 38 fresh processes on five boxes ran K1h at <= 2.30 ms (24 of 28 on four
of them; on the fifth 2 of 10, the rest at 2.35-2.52 where one block took 2.63-2.66 in all ten; two parts: 10 of 16 on
two boxes, 0 of 6 on another; eight parts average the kinds out: 2.29-2.57 ms).  Nothing is timed or probed; the
parts are plain ``torch.empty`` allocations.  (Holding tens of GB between the parts while they are allocated changes
nothing -- ``r06_hist_policy_spacer``: it is not the distance; neither does assembling ONE range from several physical
allocations, ``r06_vmm_blocks``.)
"""
import ctypes

import torch

from . import _lib

BLOCKED, INTERLEAVED = 0, 1            # MRPHY_HIST_BLOCKED / MRPHY_HIST_INTERLEAVED of include/mrphy_hip.h
MAX_PARTS = 8                          # MRPHY_HIST_MAX_PARTS

# What `sims.blochsim` does when it allocates the history itself (no workspace): `parts` parts once the history is
# at least `min_bytes` (below that a launch is too short for the placement modes to show), dealt in `layout`.
policy = {'parts': 4, 'layout': BLOCKED, 'min_bytes': 256 << 20, 'min_tiles_per_part': 8}


def set_policy(parts: int = None, layout: int = None, min_bytes: int = None, min_tiles_per_part: int = None) -> dict:
    r"""Process-wide default for the history's parts (``parts=1`` restores the single allocation of ABI <= 4).
    Returns the policy in effect."""
    if parts is not None:
        if not 1 <= int(parts) <= MAX_PARTS:
            raise ValueError(f"history parts: 1..{MAX_PARTS}, not {parts}")
        policy['parts'] = int(parts)
    if layout is not None:
        if layout not in (BLOCKED, INTERLEAVED):
            raise ValueError("history layout: BLOCKED (0) or INTERLEAVED (1)")
        policy['layout'] = layout
    if min_bytes is not None:
        policy['min_bytes'] = int(min_bytes)
    if min_tiles_per_part is not None:
        policy['min_tiles_per_part'] = max(1, int(min_tiles_per_part))
    return dict(policy)


class Hist:
    r"""``parts`` (list of 1-D tensors of the data dtype, each at least ``part_elems`` long) + ``layout``: what the
    ``_parts`` entry points take.  The ctypes pointer table is built per call (the library reads it during the call)."""
    __slots__ = ('parts', 'layout')

    def __init__(self, parts, layout=BLOCKED):
        self.parts, self.layout = list(parts), layout
        assert 1 <= len(self.parts) <= MAX_PARTS

    @property
    def device(self):
        return self.parts[0].device

    @property
    def dtype(self):
        return self.parts[0].dtype

    def c_args(self):
        r"""``(hist_parts, n_parts, layout)`` of ``mrphy_blochsim_fwd_parts`` / ``_bwd_parts``."""
        n = len(self.parts)
        return (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.parts]), n, self.layout

    def tensors(self):
        return tuple(self.parts)


def part_elems(code: int, N: int, nM: int, nT: int, n_parts: int, esize: int) -> int:
    lib = _lib.require_library()
    return max(int(lib.mrphy_blochsim_hist_part_bytes(code, N, nM, nT, n_parts)), 16) // esize


def n_parts_for(code: int, N: int, nM: int, nT: int) -> int:
    r"""The policy's part count for a history of this size (1 below ``min_bytes`` or with fewer than
    ``min_tiles_per_part`` 64-spin tiles per part)."""
    lib = _lib.require_library()
    total = int(lib.mrphy_blochsim_hist_bytes(code, N, nM, nT))
    n = policy['parts']
    tiles = (N * nM + 63) // 64
    if n <= 1 or total < policy['min_bytes'] or tiles < policy['min_tiles_per_part'] * n:
        return 1
    return n


def allocate(code: int, N: int, nM: int, nT: int, dtype, device) -> 'Hist':
    r"""The allocator's route: the policy's number of parts, each its own ``torch.empty`` -- a block of its own in the
    caching allocator, i.e. a separate ``hipMalloc`` for anything beyond a few MB, and the same blocks again on every
    later iteration of a loop (the allocator's cache)."""
    n = n_parts_for(code, N, nM, nT)
    esize = 8 if dtype == torch.float64 else 4
    pe = part_elems(code, N, nM, nT, n, esize)
    return Hist([torch.empty(pe, dtype=dtype, device=device) for _ in range(n)], policy['layout'])
