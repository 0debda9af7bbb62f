r"""A placement-aware block for ``Beff``.

``rfgr2beff`` (K0) writes ``Beff`` and ``blochsim`` (K1) reads it back: 12 B per spin-step each way, the
whole cost of the materialised route.  How fast a given allocation can be written and read is a property of
the *physical* memory behind it ("Placement"): of six 12.9-GB blocks allocated one after the
other in one process on an MI355X, four are written by K0 at 6.9 TB/s and two at 6.0 (``profiles/
r04_block_probe.json``; a slow-to-write block is read a little faster), reproducibly, whatever the kernel
does.  What the caching allocator hands out is a lottery ticket -- and it is kept for the life of the process.

:class:`BeffArena` draws a few tickets instead of one: it allocates ``candidates`` blocks (as many as the free
memory allows), times the caller's own step -- ``probe(block)``, typically ``rfgr2beff(..., out=block)`` followed
by ``blochsim(M0, block, ...)`` -- on each, keeps the fastest and releases the rest.  The block is then passed
as ``out=`` to every ``rfgr2beff`` call (an extension of the reference signature).  The reference semantics
("every call returns a fresh tensor") are the caller's to give up: the arena is for loops that consume ``Beff``
before they produce the next one, such as one rank's step of a sharded simulation (``bench.py``).
"""
from typing import Callable, Optional, Sequence

import torch

__all__ = ['BeffArena']


class BeffArena:
    r"""``arena = BeffArena(shape, dtype, device, probe)``; ``arena.block`` is the tensor to pass as ``out=``.

    Inputs:
        - ``shape``, ``dtype``, ``device``: of ``Beff``, `(N, *Nd, nT, xyz)`.
        - ``probe``: ``probe(block)`` launches the step to be timed on ``block`` (any kernels, current stream).
    Optionals:
        - ``candidates``: blocks to try (default 3; fewer if they do not fit next to ``reserve`` bytes).
        - ``reps``: timed launches per block after one untimed launch (the minimum counts).
        - ``reserve``: bytes of device memory to leave free while the candidates coexist.
    Attributes: ``block``; ``report`` -- ``{'candidate_ms': [...], 'chosen': i, 'ptr': [...]}``.
    """

    def __init__(self, shape: Sequence[int], dtype: torch.dtype, device: torch.device,
                 probe: Optional[Callable[[torch.Tensor], None]] = None, *, candidates: int = 3,
                 reps: int = 2, reserve: int = 8 << 30):
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError("BeffArena: device memory only (there is no CPU path)")
        nbytes = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            nbytes *= int(d)
        with torch.cuda.device(device):
            free, _ = torch.cuda.mem_get_info()
            free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        fit = int((free - reserve) // max(nbytes, 1))
        n = max(1, min(int(candidates), fit)) if probe is not None else 1
        blocks = [torch.empty(tuple(shape), dtype=dtype, device=device) for _ in range(n)]
        times = []
        if probe is not None and n > 1:
            ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
            with torch.cuda.device(device), torch.no_grad():
                for b in blocks:
                    probe(b)
                    best = float('inf')
                    for _ in range(max(1, reps)):
                        e0, e1 = ev(), ev()
                        e0.record()
                        probe(b)
                        e1.record()
                        e1.synchronize()
                        best = min(best, e0.elapsed_time(e1))
                    times.append(best)
            chosen = min(range(n), key=times.__getitem__)
        else:
            chosen = 0
        self.report = {'candidate_ms': [round(t, 4) for t in times], 'chosen': chosen,
                       'ptr': [hex(b.data_ptr()) for b in blocks], 'bytes_per_block': nbytes}
        self.block = blocks[chosen]
        del blocks
        if n > 1:
            torch.cuda.empty_cache()          # hand the other candidates back to the driver

    def __repr__(self):
        return f"BeffArena(shape={tuple(self.block.shape)}, {self.report})"
