r"""A placement-aware block for ``Beff``.

``rfgr2beff`` (K0) writes ``Beff`` and ``blochsim`` (K1) reads it back: 12 B per spin-step each way, the
whole cost of the materialised route.  How fast a given allocation can be written and read is a property of
the *physical* memory behind it: of six 12.9-GB blocks allocated one after the
other in one process on an MI355X, four are written by K0 at 6.9 TB/s and two at 6.0 (``profiles/
r04_block_probe.json``; a slow-to-write block is read a little faster), reproducibly, whatever the kernel
does.  What the caching allocator hands out is a lottery ticket -- and it is kept for the life of the process.

K0's store policy is a second lottery ("K0's store policy"; docs/LABNOTES.md "K1 right behind K0"): on some boxes the ``nt`` stores leave a
state in the memory-side cache that slows the ``blochsim`` that follows by 20 %, on others they do not and are the
cheaper encoding by 0-9 % of K0; ``sc1 nt`` is safe everywhere and is what ``rfgr2beff`` picks by itself below 8 GB of
``Beff`` (``nt`` from there up: ``profiles/r05_k0_store_policy.json``).

:class:`BeffArena` draws a few tickets instead of one: it allocates ``candidates`` blocks (as many as the free
memory allows), times the caller's own step -- ``probe(block)``, typically ``rfgr2beff(..., out=block)`` followed
by ``blochsim(M0, block, ...)`` -- on each, keeps the fastest and releases the rest.  The block is then passed
as ``out=`` to every ``rfgr2beff`` call (an extension of the reference signature).  The reference semantics
("every call returns a fresh tensor") are the caller's to give up: the arena is for loops that consume ``Beff``
before they produce the next one, such as one rank's step of a sharded simulation (``bench.py``).
A probe that takes a second argument, ``probe(block, store)``, is timed under both store policies per block
(``rfgr2beff(..., out=block, store=store)``); ``arena.store`` is then the faster one, to be passed on likewise.
"""
import contextvars
import inspect
from typing import Callable, Optional, Sequence

import torch

__all__ = ['BeffArena', 'GradWorkspace', 'active', 'auto']


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
        # hand the other candidates back to the driver: nothing but `self.block` may still refer to one of them (the
        # probing loop's own variable did in round 4 -- a whole un-probed block stayed reserved: ADVICE r4)
        b = None  # noqa: F841
        del blocks, b
        if n > 1:
            torch.cuda.empty_cache()

    def __repr__(self):
        return f"BeffArena(shape={tuple(self.block.shape)}, {self.report})"


# ---------------------------------------------------------------------------------------------------------------
# The gradient route: the same lottery for the blocks the history-saving forward (K1h) and the adjoint sweep (K3)
# WRITE.  DESIGN.md §4: K1h runs at 0.61-0.62 or 0.74-0.75 of HBM peak depending on the physical memory
# behind the history block, K3 at 0.60 or 0.70-0.74 depending on the one behind grad_Beff, independently of each
# other, whatever the kernels do (profiles/r03_placement_vs_size.json); the blocks `sims.blochsim` draws from the
# caching allocator are kept for the life of the process.
# ---------------------------------------------------------------------------------------------------------------
_ACTIVE = contextvars.ContextVar('mrphy_amd_grad_workspace', default=None)


def active(shape=None, dtype=None, device=None):
    r"""The :class:`GradWorkspace` of the enclosing ``with ws:`` block of this thread / context -- or, inside a
    ``with workspace.auto():`` block, the pool's workspace for this ``Beff`` shape (built and probed at first use) -- or
    ``None``."""
    w = _ACTIVE.get()
    if isinstance(w, auto):
        return None if shape is None else w.get(shape, dtype, device)
    return w


class auto:
    r"""``with mrphy_amd.workspace.auto(): ...`` -- every ``sims.blochsim`` call of this thread / context that needs a
    gradient draws its history and ``grad_Beff`` from a :class:`GradWorkspace` of its own ``Beff`` shape, built (and
    placement-probed) the first time that shape is seen and kept by this object: the reference-signature gradient route
    (``rfgr2beff`` -> ``blochsim`` -> ``backward``, e.g. ``mobjs`` with ``install(fuse_applypulse=False)``) gets the
    probed blocks without the caller knowing the shapes.  The workspaces' trade applies (one forward / backward pair in
    flight per shape, ``grad_Beff`` storage reused from one backward to the next): an opt-in, like ``workspace=``."""

    def __init__(self, candidates: int = 24, reserve: int = 8 << 30):
        self.candidates, self.reserve = candidates, reserve
        self.pool = {}
        self._tokens = []

    def get(self, shape, dtype, device):
        key = (tuple(int(d) for d in shape), dtype, str(device))
        ws = self.pool.get(key)
        if ws is None:
            tok = _ACTIVE.set(None)           # the probe's own blochsim calls pass their blocks explicitly
            try:
                ws = self.pool[key] = GradWorkspace(shape, dtype, device, candidates=self.candidates,
                                                    reserve=self.reserve, with_beff=False)
            finally:
                _ACTIVE.reset(tok)
        return ws

    def __enter__(self):
        self._tokens.append(_ACTIVE.set(self))
        return self

    def __exit__(self, *exc):
        _ACTIVE.reset(self._tokens.pop())
        return False


class _Pair:
    r"""What ``sims.BlochSimHIP`` draws from while the workspace probes: one candidate assignment (no guard)."""
    generation = 0

    def __init__(self, hist, grad):
        self._hist, self._grad = hist, grad

    def take_hist(self, elems, dtype, device=None):
        return self._hist[:elems]

    def take_grad(self, shape, dtype, generation):
        n = 1
        for d in shape:
            n *= int(d)
        return self._grad[:n].view(tuple(shape))


class GradWorkspace:
    r"""Placement-probed blocks for ``sims.blochsim`` + ``backward`` over a materialised ``Beff`` of one shape:
    the history the forward writes (``sims.py:84-88`` of the reference, 12 instead of 40 B per spin-step here), the
    ``grad_Beff`` the adjoint writes (``sims.py:239-264``) and, optionally, the ``Beff`` block itself.

    ``ws = GradWorkspace(beff_shape, dtype, device)`` draws candidate blocks one after the other -- up to
    ``candidates``, as many as fit beside ``reserve`` bytes -- and times K1h with each as its history and K3 with each
    as its ``grad_Beff`` (the library's own kernels on a synthetic field: the rates are a property of the memory, not
    of the data).  A block is fast or slow for BOTH kernels, by 20 %, and how many blocks are of the fast kind is the
    box's and the process's lottery: one 6.4-GB block in five to eight on three boxes, every block on a fourth; two
    25.8-GB blocks in four or five (``profiles/r05_grad_workspace.json``).  What makes a block fast is where its
    physical pages lie, which the driver decides: a 6-GiB window sliding through ONE 64-GiB allocation is slow
    everywhere except within +-3 GiB of the allocation's 32-GiB mark, fastest when the mark is at its centre
    (``profiles/r05_placement_windows_64c_x2048.json``).
    So the draw goes on until two blocks are within 4 % of the best seen while, for each of the two kernels, a clearly
    slower one (> 10 %) shows that the best is the fast mode -- or the candidates are used up (24 by default: transient
    memory, 0.03 s of probing each at 64^3 x 2048).  The history and ``grad_Beff`` get the pair with the smallest
    K1h + K3, the rest goes back to the driver.  Then::

        beff = rfgr2beff(rf, gr, loc, ..., out=ws.beff)               # optional (with_beff=True)
        Mo = sims.blochsim(Mi, beff, T1=..., T2=..., workspace=ws)    # or:  with ws: cube.applypulse(...)
        Mo.sum().backward()                                           # grad_Beff is ws's block

    Extension of the reference signature, with the arena's trade: ONE forward / backward pair is in flight per
    workspace -- a second forward overwrites the history of the first (its backward then raises instead of
    differentiating the wrong trajectory), and every backward returns the same ``grad_Beff`` storage.  Results are
    bit-identical to the allocator's route (the kernels do not know where their blocks came from).

    Attributes: ``beff`` (or ``None``), ``report`` -- ``{'K1h_ms': [...], 'K3_ms': [...], 'chosen': {'hist': i,
    'grad': j}, 'probed': bool, 'stopped': why}``.
    """

    _MIN_PROBE_BYTES = 64 << 20       # below this the launch overhead hides the difference: nothing is probed

    def __init__(self, shape: Sequence[int], dtype: torch.dtype, device: torch.device, *, candidates: int = 24,
                 reps: int = 2, reserve: int = 8 << 30, with_beff: bool = True):
        from . import _lib
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError("GradWorkspace: device memory only (there is no CPU path)")
        if dtype not in (torch.float32, torch.float64):
            raise NotImplementedError(f"GradWorkspace: {dtype}; float32 and float64 are implemented")
        shape = tuple(int(d) for d in shape)
        assert len(shape) >= 4 and shape[-1] == 3, "GradWorkspace: shape of Beff, (N, *Nd, nT, xyz)"
        lib = _lib.require_library()
        N, nT = shape[0], shape[-2]
        nM = 1
        for d in shape[1:-2]:
            nM *= d
        esz = 8 if dtype == torch.float64 else 4
        code = _lib.F64 if dtype == torch.float64 else _lib.F32P
        self.shape, self.dtype, self.device = shape, dtype, device
        self._numel = N * nM * nT * 3
        self._hist_elems = max(int(lib.mrphy_blochsim_hist_bytes(code, N, nM, nT)), 16) // esz
        block_elems = max(self._numel, self._hist_elems, 4)
        nbytes = block_elems * esz
        with torch.cuda.device(device):
            free, _ = torch.cuda.mem_get_info()
            free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        fit = int((free - reserve) // nbytes) - 1            # one block holds the field the probe reads (= Beff's)
        P = max(2, min(int(candidates), fit))
        new = lambda: torch.empty(block_elems, dtype=dtype, device=device)  # noqa: E731
        field = new()                                        # the probe's field; the Beff block if one is wanted
        blocks, tH, tG, why = [new(), new()], [], [], 'not probed (small blocks or no spare memory)'
        probe = nbytes >= self._MIN_PROBE_BYTES and P > 2
        if probe:
            why = self._probe(field, blocks, tH, tG, new, P, N, nM, nT, reps)
            _, h, g = min(((tH[h] + tG[g], h, g) for h in range(len(blocks)) for g in range(len(blocks)) if g != h))
        else:
            h, g = 0, 1
        self.report = {'K1h_ms': [round(t, 4) for t in tH], 'K3_ms': [round(t, 4) for t in tG],
                       'chosen': {'hist': h, 'grad': g}, 'probed': bool(probe), 'stopped': why,
                       'ptr': [hex(b.data_ptr()) for b in blocks], 'bytes_per_block': nbytes}
        self._hist, self._grad = blocks[h], blocks[g]
        self._field = field if with_beff else None
        self.beff = field[:self._numel].view(shape) if with_beff else None
        n_drawn = len(blocks)
        del blocks, field, new
        if n_drawn > 2 or not with_beff:
            torch.cuda.empty_cache()          # the candidates that lost go back to the driver
        self.generation = 0
        self._tokens = []

    # -- probing ---------------------------------------------------------------------------------------------
    def _probe(self, field, blocks, tH, tG, new, P, N, nM, nT, reps):
        r"""Times K1h writing its history into each candidate and K3 writing ``grad_Beff`` into it (reading its history
        from the candidate drawn before), appending candidates to ``blocks`` until the stopping rule of the class
        docstring holds or ``P`` are drawn.  The field is smooth noise of realistic size (up to 0.37 rad per step).
        (The probe runs WITHOUT relaxation and with ``γ`` / ``dt`` at the reference's fp64 defaults, so its launches are
        other instances of the kernels than a caller's -- ``<prec_f64, false, ...>`` / ``<double, false, ...>`` in a profile
        -- with the same memory traffic: in ``rocprofv3 --stats`` the caller's rows stay free of the probe's launches.)"""
        from . import sims
        dev, dtype = self.device, self.dtype
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

        def timed(fn):
            fn()
            best = float('inf')
            for _ in range(max(1, reps)):
                e0, e1 = ev(), ev()
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best

        Mi = torch.zeros((N, nM, 3), dtype=dtype, device=dev)
        Mi[..., 2] = 1
        kw = {}                  # no relaxation, default γ / dt: see the note on kernel instances above
        gMo = torch.ones_like(Mi)
        beff = field[:self._numel].view(self.shape)
        beff.uniform_(-2.0, 2.0)                               # Gauss: |γ2πdt B| up to 0.37 rad with γH, dt0
        beff.requires_grad_(True)

        def measure(i):
            other = blocks[i - 1] if i > 0 else blocks[1]
            tH.append(timed(lambda: sims.blochsim(Mi, beff, workspace=_Pair(blocks[i], other), **kw)))
            Mo = sims.blochsim(Mi, beff, workspace=_Pair(other, blocks[i]), **kw)
            tG.append(timed(lambda: torch.autograd.grad(Mo, beff, gMo, retain_graph=True)))

        def settled():
            mH, mG = min(tH), min(tG)
            fH = [i for i, t in enumerate(tH) if t <= 1.04 * mH]
            fG = [i for i, t in enumerate(tG) if t <= 1.04 * mG]
            pair = any(h != g for h in fH for g in fG)
            # ... and, for EACH kernel, a clearly slower block that shows its best is the fast mode (round-5 run: two
            # candidates at K3 3.69 / 3.72 ms -- both slow -- passed an `or` here on the strength of K1h's spread alone)
            return pair and max(tH) >= 1.10 * mH and max(tG) >= 1.10 * mG

        with torch.cuda.device(dev):
            measure(0)
            measure(1)
            while True:
                if settled():
                    return 'two blocks of the fast kind found'
                if len(blocks) >= P:
                    return 'candidates used up'
                blocks.append(new())
                measure(len(blocks) - 1)

    # -- what sims.BlochSimHIP draws ----------------------------------------------------------------------------
    def take_hist(self, elems: int, dtype: torch.dtype, device=None):
        if device is not None and torch.device(device) != self.device:
            raise RuntimeError(f"GradWorkspace built on {self.device}: this call's tensors are on {device}")
        if dtype != self.dtype or elems > self._hist.numel():
            raise RuntimeError(f"GradWorkspace built for Beff {self.shape} {self.dtype}: this call needs a history of "
                               f"{elems} {dtype} elements")
        self.generation += 1
        return self._hist[:elems]

    def take_grad(self, shape, dtype: torch.dtype, generation: int):
        if generation != self.generation:
            raise RuntimeError("GradWorkspace: the history of this forward has been overwritten by a later "
                               "sims.blochsim(..., workspace=) call on the same workspace (one forward / backward "
                               "pair in flight per workspace)")
        n = 1
        for d in shape:
            n *= int(d)
        if dtype != self.dtype or n > self._grad.numel():
            raise RuntimeError(f"GradWorkspace built for Beff {self.shape} {self.dtype}: grad_Beff {tuple(shape)} {dtype}")
        return self._grad[:n].view(tuple(shape))

    # -- `with ws:` makes it the default of sims.blochsim in this thread / context ---------------------------------
    def __enter__(self):
        self._tokens.append(_ACTIVE.set(self))
        return self

    def __exit__(self, *exc):
        _ACTIVE.reset(self._tokens.pop())
        return False

    def __repr__(self):
        return f"GradWorkspace(shape={self.shape}, {self.report})"
