r"""A placement-aware block for ``Beff``.

``rfgr2beff`` (K0) writes ``Beff`` and ``blochsim`` (K1) reads it back: 12 B per spin-step each way, the
whole cost of the materialised route.  How fast a given allocation can be written and read is a property of
the *physical* memory behind it ("Placement"): of six 12.9-GB blocks allocated one after the
other in one process on an MI355X, four are written by K0 at 6.9 TB/s and two at 6.0 (``profiles/
r04_block_probe.json``; a slow-to-write block is read a little faster), reproducibly, whatever the kernel
does.  What the caching allocator hands out is a lottery ticket -- and it is kept for the life of the process.

K0's store policy is a second lottery ("K1 right behind K0"): on some boxes the ``nt`` stores leave a
state in the memory-side cache that slows the ``blochsim`` that follows by 20 %, on others they do not and are the
cheaper encoding by 0-9 % of K0; ``sc1 nt`` is safe everywhere and is what ``rfgr2beff`` picks by itself.

:class:`BeffArena` draws a few tickets instead of one: it allocates ``candidates`` blocks (as many as the free
memory allows), times the caller's own step -- ``probe(block)``, typically ``rfgr2beff(..., out=block)`` followed
by ``blochsim(M0, block, ...)`` -- on each, keeps the fastest and releases the rest.  The block is then passed
as ``out=`` to every ``rfgr2beff`` call (an extension of the reference signature).  The reference semantics
("every call returns a fresh tensor") are the caller's to give up: the arena is for loops that consume ``Beff``
before they produce the next one, such as one rank's step of a sharded simulation (``bench.py``).
A probe that takes a second argument, ``probe(block, store)``, is timed under both store policies per block
(``rfgr2beff(..., out=block, store=store)``); ``arena.store`` is then the faster one, to be passed on likewise.
"""
import inspect
from typing import Callable, Optional, Sequence

import torch

__all__ = ['BeffArena']


class BeffArena:
    r"""``arena = BeffArena(shape, dtype, device, probe)``; ``arena.block`` is the tensor to pass as ``out=``.

    Inputs:
        - ``shape``, ``dtype``, ``device``: of ``Beff``, `(N, *Nd, nT, xyz)`.
        - ``probe``: ``probe(block)`` -- or ``probe(block, store)`` -- launches the step to be timed on ``block``
          (any kernels, current stream).
    Optionals:
        - ``candidates``: blocks to try (default 3; fewer if they do not fit next to ``reserve`` bytes).
        - ``reps``: timed launches per block after one untimed launch (the minimum counts).
        - ``reserve``: bytes of device memory to leave free while the candidates coexist.
        - ``stores``: the policies a two-argument probe is timed with (default ``('sc1nt', 'nt')``: the first is kept
          unless a later one is more than 0.5 % faster on the chosen block).
    Attributes: ``block``; ``store`` (``None`` with a one-argument probe); ``report`` -- ``{'candidate_ms': [...],
    'chosen': i, 'ptr': [...]}`` and, with policies, ``'by_store': {policy: [...]}``, ``'store': policy``.
    """

    def __init__(self, shape: Sequence[int], dtype: torch.dtype, device: torch.device,
                 probe: Optional[Callable[..., None]] = None, *, candidates: int = 3,
                 reps: int = 2, reserve: int = 8 << 30, stores: Sequence[str] = ('sc1nt', 'nt')):
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
        with_store = probe is not None and len(inspect.signature(probe).parameters) >= 2
        pols = tuple(stores) if with_store else (None,)
        by_store = {pol: [] for pol in pols}
        times = []
        self.store = None
        if probe is not None and (n > 1 or len(pols) > 1):
            ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
            call = (lambda b, pol: probe(b, pol)) if with_store else (lambda b, pol: probe(b))
            with torch.cuda.device(device), torch.no_grad():
                for b in blocks:
                    call(b, pols[0])              # first touch of the block: not the first policy's to pay
                    for pol in pols:
                        call(b, pol)
                        best = float('inf')
                        for _ in range(max(1, reps)):
                            e0, e1 = ev(), ev()
                            e0.record()
                            call(b, pol)
                            e1.record()
                            e1.synchronize()
                            best = min(best, e0.elapsed_time(e1))
                        by_store[pol].append(best)
            times = [min(by_store[pol][i] for pol in pols) for i in range(n)]
            chosen = min(range(n), key=times.__getitem__)
            if with_store:                    # the first policy of `stores` unless another one beats it by more than 0.5 %
                self.store = pols[0]
                for pol in pols[1:]:
                    if by_store[pol][chosen] < 0.995 * by_store[self.store][chosen]:
                        self.store = pol
        else:
            chosen = 0
        self.report = {'candidate_ms': [round(t, 4) for t in times], 'chosen': chosen,
                       'ptr': [hex(b.data_ptr()) for b in blocks], 'bytes_per_block': nbytes}
        if with_store:
            self.report['by_store'] = {pol: [round(t, 4) for t in ts] for pol, ts in by_store.items()}
            self.report['store'] = self.store
        self.block = blocks[chosen]
        del blocks
        if n > 1:
            torch.cuda.empty_cache()          # hand the other candidates back to the driver

    def __repr__(self):
        return f"BeffArena(shape={tuple(self.block.shape)}, {self.report})"
