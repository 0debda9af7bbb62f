r"""mrphy_amd -- MI355X-native (gfx950) drop-in for the Bloch-simulation hot path of MRphy.py.

Only the hot path is here (reference: tianrluo/MRphy.py v0.2.0):

=====================================  =====================================================
this package                            reference it replaces
=====================================  =====================================================
``mrphy_amd.beffective.rfgr2beff``      ``mrphy/beffective.py:107-168``
``mrphy_amd.sims.blochsim``             ``mrphy/sims.py:272-315`` (+ ``BlochSim`` fwd/bwd)
``mrphy_amd.slowsims.blochsim_1step``   ``mrphy/slowsims.py:15-54``
``mrphy_amd.sims.freeprec``             ``mrphy/sims.py:424-458`` (+ ``FreePrec`` fwd/bwd)
``mrphy_amd.interp.interpT``            ``mobjs.Pulse.interpT`` (linear), ``mobjs.py:177-220``
``mrphy_amd.masks.extract/embed``       ``mobjs.SpinArray.extract/embed``, ``mobjs.py:512-553``
``mrphy_amd.masks.cube_loc``            ``mobjs.SpinCube._update_loc_``, ``mobjs.py:815-839``
``mrphy_amd.beffective.beff2ab``        ``mrphy/beffective.py:40-104``
``mrphy_amd.slowsims.blochsim_ab``      ``mrphy/slowsims.py:117-131``
=====================================  =====================================================

Every function keeps the reference's name, keyword names (``γ``, ``Δf`` ...), tensor layouts
and error behaviour, takes PyTorch-ROCm tensors and launches hand-written HIP kernels through
the C ABI of ``libmrphy_hip.so`` (``include/mrphy_hip.h``) on torch's current stream.  There is
NO CPU fallback in these functions: CPU tensors or a missing library raise.  (After ``install()`` into a
reference ``mrphy``, all-CPU calls are handed back to that ``mrphy``'s own functions -- the user's code, not ours.)

``install()`` swaps these functions (and the ``mobjs`` methods either side of the path:
``SpinArray.extract/embed``, ``SpinCube._update_loc_``, ``Pulse.interpT``; ``SpinArray.applypulse``
runs its two calls as the one fused kernel) into an importable ``mrphy`` so that
``mrphy.mobjs.SpinArray/SpinCube/Pulse`` run on this path unchanged.

The directory is called ``mrphy.py_amd``; import it as ``mrphy_amd`` (see ``mrphy_amd.py`` at
the repository root).
"""
from math import pi as π, inf  # noqa: F401

from ._consts import γH, T1G, T2G, dt0, gmax0, smax0, rfmax0  # noqa: F401

__version__ = '0.1.0'

from . import _lib, beffective, sims, slowsims, utils, fused, interp, masks, synth, dist, workspace  # noqa: E402,F401
from ._lib import build, library_path, require_library  # noqa: E402,F401
from ._host import constants_on, precision  # noqa: E402,F401

__all__ = ['γH', 'T1G', 'T2G', 'dt0', 'gmax0', 'smax0', 'rfmax0', 'π',
           'beffective', 'sims', 'slowsims', 'utils', 'fused', 'interp', 'masks', 'synth', 'dist',
           'build', 'install', 'uninstall', 'constants_on', 'precision']

_saved = {}
_mask_index = None      # WeakIdKeyDictionary: SpinArray.mask tensor (by identity) -> masks.MaskIndex


def _index_of(mask):
    r"""The :class:`masks.MaskIndex` of a ``SpinArray.mask``, built once per mask tensor
    (``mobjs.py:274``: masks are not to be modified)."""
    global _mask_index
    if _mask_index is None:
        # keyed by IDENTITY: weakref.WeakKeyDictionary compares keys with ==, and Tensor.__eq__
        # is elementwise (bool() of the result raises for a multi-element mask)
        from torch.utils.weak import WeakIdKeyDictionary
        _mask_index = WeakIdKeyDictionary()
    ix = _mask_index.get(mask)
    if ix is None:
        ix = _mask_index[mask] = masks.MaskIndex(mask)
    return ix


# Object-layer glue for install().  mobjs builds its objects on the CPU by default and moves them
# with ``.to(device=...)`` (which constructs a new object, mobjs.py:667-685,946-965), so an object
# that lives on the CPU keeps the reference's own methods; one that lives on the ROCm device uses
# the kernels.  (The functions of :mod:`mrphy_amd.masks` themselves have no CPU path.)
def _spinarray_extract(self, v, *, out_=None):
    if self.device.type != 'cuda':
        return _saved['extract'](self, v, out_=out_)
    return masks.extract(v, _index_of(self.mask), out_=out_)


def _spinarray_embed(self, v_, *, out=None):
    if self.device.type != 'cuda':
        return _saved['embed'](self, v_, out=out)
    return masks.embed(v_, _index_of(self.mask), out=out)


def _spincube_update_loc_(self):
    if self.spinarray.device.type != 'cuda':
        return _saved['_update_loc_'](self)
    masks.cube_loc(_index_of(self.spinarray.mask), self.fov, self.ofst, out_=self.loc_)
    return


def _spinarray_applypulse(self, pulse, *, doEmbed: bool = False, doRelax: bool = True,
                          doUpdate: bool = False, loc=None, loc_=None, Δf=None, Δf_=None,
                          b1Map=None, b1Map_=None):
    r"""``mobjs.SpinArray.applypulse`` (``mobjs.py:394-450``) for a device-resident spin array: the
    same argument checks, the same gathers of spatial ``loc / Δf / b1Map`` into the compact layout,
    the same ``doRelax / doUpdate / doEmbed`` handling -- but the ``pulse2beff`` + ``sims.blochsim``
    pair in the middle (``mobjs.py:435-446``: ``Pulse.beff`` -> ``rfgr2beff`` -> a temporary
    ``(N, nM, nT, xyz)`` tensor that nothing else ever sees -> ``blochsim``) runs as ONE kernel,
    :func:`mrphy_amd.fused.blochsim_rfgr`: the result is bit-identical (the fused kernel equals the
    two-kernel path bit for bit), the 12 B per spin and step never go to HBM, and a backward pass to
    the pulse uses the fused adjoint.  ``SpinCube.applypulse`` (``mobjs.py:840-869``) calls this method
    and inherits it.  CPU arrays go to the reference's own method."""
    if self.device.type != 'cuda':
        return _saved['applypulse'](self, pulse, doEmbed=doEmbed, doRelax=doRelax, doUpdate=doUpdate,
                                    loc=loc, loc_=loc_, Δf=Δf, Δf_=Δf_, b1Map=b1Map, b1Map_=b1Map_)
    # the reference's argument rules (mobjs.py:425-433): exactly one of loc / loc_; at most one of each
    # spatial / compact pair; a spatial map is gathered through the mask
    assert (loc is None) ^ (loc_ is None)
    assert Δf is None or Δf_ is None
    assert b1Map is None or b1Map_ is None
    gather = lambda spatial, compact: compact if spatial is None else self.extract(spatial)  # noqa: E731
    loc_, Δf_, b1Map_ = gather(loc, loc_), gather(Δf, Δf_), gather(b1Map, b1Map_)
    # SpinArray.pulse2beff (mobjs.py:651-653) moves the pulse to the array's device and dtype, and
    # Pulse.beff (mobjs.py:167-170) the maps to the pulse's device; blochsim gets the ORIGINAL pulse's dt
    p = pulse.to(device=self.device, dtype=self.dtype)
    on = lambda x: None if x is None else x.to(device=p.device)  # noqa: E731
    T1, T2 = (self.T1_, self.T2_) if doRelax else (None, None)
    M_ = fused.blochsim_rfgr(self.M_, p.rf, p.gr, on(loc_), Δf=on(Δf_), b1Map=on(b1Map_),
                             γ_beff=on(self.γ_), T1=T1, T2=T2, γ=self.γ_, dt=pulse.dt)
    if doUpdate:
        self.M_ = M_
    return self.embed(M_) if doEmbed else M_


_INTERP_GRAPH = False


# Function targets of install().  The kernels' host layer has no CPU path and raises for a CPU tensor -- on purpose,
# and it stays that way for anyone who calls ``mrphy_amd.sims.blochsim`` directly.  But ``install()`` replaces the
# functions of the USER'S OWN ``mrphy``, whose objects live on the CPU by default (mobjs.py:85,268): a CPU
# ``SpinCube.applypulse`` must keep working after ``install()`` exactly as before it (SURVEY §8b "Fallback";
# BASELINE configs[0] is a CPU case).  So what is installed is a router per function: a call whose tensors are all on
# the CPU goes to the saved reference callable -- the user's own code, bit for bit what it computed before; a call
# with any tensor elsewhere goes to the HIP path (and never reaches the reference).  The five methods above route
# the same way, by the object's device.  Nothing of ``oracle/`` is involved.
_routed = {}


def _all_on_cpu(args, kwargs) -> bool:
    from torch import Tensor
    ts = [a for a in list(args) + list(kwargs.values()) if isinstance(a, Tensor)]
    return bool(ts) and all(t.device.type == 'cpu' for t in ts)


def _route(name, hip_fn):
    r"""``hip_fn`` for device tensors, ``_saved[name]`` (the reference's own function) for all-CPU calls."""
    import functools

    @functools.wraps(hip_fn)
    def routed(*args, **kwargs):
        if _all_on_cpu(args, kwargs):
            return _saved[name](*args, **kwargs)
        return hip_fn(*args, **kwargs)
    routed.hip = hip_fn
    _routed[name] = routed
    return routed


def _pulse_interpT(self, dt, *, kind: str = 'linear'):
    r"""``mobjs.Pulse.interpT`` (``mobjs.py:177-220``) for a device-resident pulse: same asserts,
    same early ``deepcopy`` for an unchanged dwell time, same ``desc``, a new ``Pulse`` on the
    pulse's device/dtype that does NOT carry ``gmax/smax/rfmax`` over (``mobjs.py:219-220``) --
    but the waveforms are resampled by :func:`mrphy_amd.interp.interpT` on the device instead of
    ``detach().cpu().numpy()`` -> scipy -> device.  The new waveforms are detached leaves, as the
    reference's are (``mobjs.py:203``), unless ``install(interpT_graph=True)`` asked for the
    differentiable form.  CPU pulses go to the reference's own method."""
    if self.device.type != 'cuda':
        return _saved['interpT'](self, dt, kind=kind)
    assert (self.dt.numel() == dt.numel() == 1)
    dt_o, dt_n = self.dt.item(), dt.item()
    if dt_o == dt_n:
        import copy
        return copy.deepcopy(self)
    rf, gr = (self.rf, self.gr) if _INTERP_GRAPH else (self.rf.detach(), self.gr.detach())
    rf_n, gr_n, _ = interp.interpT(rf, gr, self.dt, dt, kind=kind)
    desc = f"{self.desc} + interpT'ed: dt = {dt_n}"
    return type(self)(rf_n, gr_n, dt=dt, desc=desc, device=self.device, dtype=self.dtype)


_AUTO_WS = None          # install(grad_workspace=True): the pool of placement-probed workspaces of the routed sims.blochsim


def _blochsim_in_pool(*args, **kwargs):
    r"""``sims.blochsim`` inside the installed :class:`workspace.auto` pool (one workspace per ``Beff`` shape)."""
    if _AUTO_WS is None or kwargs.get('workspace') is not None or workspace.active() is not None:
        return sims.blochsim(*args, **kwargs)
    # a LOCAL reset token: the pool object is shared by every thread that calls the routed function
    tok = workspace._ACTIVE.set(_AUTO_WS)
    try:
        return sims.blochsim(*args, **kwargs)
    finally:
        workspace._ACTIVE.reset(tok)


def install(mrphy=None, *, lazy_beff: bool = False, interpT_graph: bool = False,
            fuse_applypulse: bool = True, grad_workspace: bool = False):
    r"""Route an importable reference ``mrphy`` through this package.

    Replaces ``mrphy.beffective.rfgr2beff``, ``mrphy.sims.blochsim``, ``mrphy.sims.freeprec``,
    ``mrphy.slowsims.blochsim_1step`` (the call targets of ``mrphy.mobjs``, ``mobjs.py:173,446,588``),
    ``beffective.beff2ab`` and ``slowsims.blochsim_ab`` with routers (:func:`_route`): device tensors run the
    HIP-backed functions, all-CPU calls go to the reference's own functions saved here -- so CPU objects behave
    after ``install()`` exactly as before it.  ``mobjs`` looks them up as module
    attributes at call time, so ``SpinArray.applypulse`` etc. need no change.  The mask
    gather/scatter either side of them -- ``SpinArray.extract/embed`` and
    ``SpinCube._update_loc_`` (``mobjs.py:512-553,815-839``) -- are replaced by the index-list
    kernels of :mod:`mrphy_amd.masks`.

    ``mobjs.Pulse.interpT`` (``mobjs.py:177-220``, the multi-scale caller of BASELINE configs[4])
    resamples device-resident pulses with the on-device kernels (:func:`_pulse_interpT`);
    ``interpT_graph=True`` additionally keeps the autograd graph through the resampling (the
    reference cuts it at ``mobjs.py:203``).

    ``SpinArray.applypulse`` (``mobjs.py:394-450``; ``SpinCube.applypulse`` goes through it) runs
    its ``pulse2beff`` + ``blochsim`` pair as the ONE fused kernel for device-resident arrays
    (:func:`_spinarray_applypulse`: same result bit for bit, no ``(N,nM,nT,3)`` temporary, fused
    adjoint); ``fuse_applypulse=False`` leaves the method alone (the two calls then reach
    ``rfgr2beff`` and ``blochsim`` separately).

    ``grad_workspace=True`` gives the routed ``sims.blochsim`` a pool of placement-probed workspaces
    (:class:`mrphy_amd.workspace.auto`: one :class:`~mrphy_amd.workspace.GradWorkspace` per ``Beff`` shape and thread, built
    the first time a gradient is wanted at that shape): the history and ``grad_Beff`` of the reference-signature gradient
    route are then its blocks -- same bits, but NOT the reference's (or this package's default) ownership: one forward /
    backward pair is in flight per shape, and **every backward at a shape returns the same ``grad_Beff`` storage** -- a
    ``Beff.grad`` (or any tensor autograd derived from it without a copy) kept from one iteration is overwritten by the
    next iteration's backward.  That is the aliasing hazard of ``sims.py:239-264``, which ``sims.blochsim`` without a
    workspace deliberately does not have; it is hidden behind the reference signature here, which is why this flag is
    off by default.  Loops that consume their gradients before the next backward (every optimiser step does) are safe;
    code that collects ``grad_Beff`` tensors across iterations must ``clone()`` them.  The pool pins at most 64 GiB
    (least recently used shapes are dropped) and a shape whose workspace cannot be built is served by the allocator
    (DESIGN.md §4).

    ``lazy_beff=True`` makes ``rfgr2beff`` return a :class:`beffective.LazyBeff` handle that
    ``blochsim`` consumes with the fused kernel (no ``(N,nM,nT,3)`` tensor in HBM); any other
    use of the handle materialises it.
    """
    if mrphy is None:
        import mrphy  # noqa: F811  (the reference package, if importable)
    if not _saved:
        _saved['rfgr2beff'] = mrphy.beffective.rfgr2beff
        _saved['blochsim'] = mrphy.sims.blochsim
        _saved['blochsim_1step'] = mrphy.slowsims.blochsim_1step
        _saved['freeprec'] = mrphy.sims.freeprec
        _saved['beff2ab'] = mrphy.beffective.beff2ab
        _saved['blochsim_ab'] = mrphy.slowsims.blochsim_ab
        _saved['extract'] = mrphy.mobjs.SpinArray.extract
        _saved['embed'] = mrphy.mobjs.SpinArray.embed
        _saved['_update_loc_'] = mrphy.mobjs.SpinCube._update_loc_
        _saved['interpT'] = mrphy.mobjs.Pulse.interpT
        _saved['applypulse'] = mrphy.mobjs.SpinArray.applypulse
    global _INTERP_GRAPH, _AUTO_WS
    _INTERP_GRAPH = bool(interpT_graph)
    _AUTO_WS = workspace.auto() if grad_workspace else None
    beffective.LAZY_DEFAULT = bool(lazy_beff)
    mrphy.beffective.rfgr2beff = _route('rfgr2beff', beffective.rfgr2beff)
    mrphy.sims.blochsim = _route('blochsim', _blochsim_in_pool if grad_workspace else sims.blochsim)
    mrphy.slowsims.blochsim_1step = _route('blochsim_1step', slowsims.blochsim_1step)
    mrphy.sims.freeprec = _route('freeprec', sims.freeprec)          # mobjs.SpinArray.freeprec (mobjs.py:588)
    mrphy.beffective.beff2ab = _route('beff2ab', beffective.beff2ab)
    mrphy.slowsims.blochsim_ab = _route('blochsim_ab', slowsims.blochsim_ab)
    mrphy.mobjs.SpinArray.extract = _spinarray_extract
    mrphy.mobjs.SpinArray.embed = _spinarray_embed
    mrphy.mobjs.SpinCube._update_loc_ = _spincube_update_loc_
    mrphy.mobjs.Pulse.interpT = _pulse_interpT
    mrphy.mobjs.SpinArray.applypulse = _spinarray_applypulse if fuse_applypulse else _saved['applypulse']
    return mrphy


def uninstall(mrphy=None):
    r"""Undo :func:`install`."""
    if mrphy is None:
        import mrphy  # noqa: F811
    if _saved:
        mrphy.beffective.rfgr2beff = _saved.pop('rfgr2beff')
        mrphy.sims.blochsim = _saved.pop('blochsim')
        mrphy.slowsims.blochsim_1step = _saved.pop('blochsim_1step')
        mrphy.sims.freeprec = _saved.pop('freeprec')
        mrphy.beffective.beff2ab = _saved.pop('beff2ab')
        mrphy.slowsims.blochsim_ab = _saved.pop('blochsim_ab')
        mrphy.mobjs.SpinArray.extract = _saved.pop('extract')
        mrphy.mobjs.SpinArray.embed = _saved.pop('embed')
        mrphy.mobjs.SpinCube._update_loc_ = _saved.pop('_update_loc_')
        mrphy.mobjs.Pulse.interpT = _saved.pop('interpT')
        mrphy.mobjs.SpinArray.applypulse = _saved.pop('applypulse')
    _routed.clear()
    global _INTERP_GRAPH, _AUTO_WS
    _INTERP_GRAPH = False
    _AUTO_WS = None
    beffective.LAZY_DEFAULT = False
    return mrphy
