r"""mrphy_amd -- MI355X-native (gfx950) drop-in for the Bloch-simulation hot path of MRphy.py.

Only the hot path is here (reference: tianrluo/MRphy.py v0.2.0):

=====================================  =====================================================
this package                            reference it replaces
=====================================  =====================================================
``mrphy_amd.beffective.rfgr2beff``      ``mrphy/beffective.py:107-168``
``mrphy_amd.sims.blochsim``             ``mrphy/sims.py:272-315`` (+ ``BlochSim`` fwd/bwd)
``mrphy_amd.slowsims.blochsim_1step``   ``mrphy/slowsims.py:15-54``
=====================================  =====================================================

Every function keeps the reference's name, keyword names (``γ``, ``Δf`` ...), tensor layouts
and error behaviour, takes PyTorch-ROCm tensors and launches hand-written HIP kernels through
the C ABI of ``libmrphy_hip.so`` (``include/mrphy_hip.h``) on torch's current stream.  There is
NO CPU fallback: CPU tensors or a missing library raise.

``install()`` swaps the three functions into an importable ``mrphy`` so that
``mrphy.mobjs.SpinArray/SpinCube/Pulse`` run on this path unchanged.

The directory is called ``mrphy.py_amd``; import it as ``mrphy_amd`` (see ``mrphy_amd.py`` at
the repository root).
"""
from math import pi as π, inf  # noqa: F401

from ._consts import γH, T1G, T2G, dt0, gmax0, smax0, rfmax0  # noqa: F401

__version__ = '0.1.0'

from . import _lib, beffective, sims, slowsims, utils, fused, interp, synth, dist  # noqa: E402,F401
from ._lib import build, library_path, require_library  # noqa: E402,F401
from ._host import constants_on  # noqa: E402,F401

__all__ = ['γH', 'T1G', 'T2G', 'dt0', 'gmax0', 'smax0', 'rfmax0', 'π',
           'beffective', 'sims', 'slowsims', 'utils', 'fused', 'interp', 'synth', 'dist',
           'build', 'install', 'uninstall', 'constants_on']

_saved = {}


def install(mrphy=None, *, lazy_beff: bool = False):
    r"""Route an importable reference ``mrphy`` through this package.

    Replaces ``mrphy.beffective.rfgr2beff``, ``mrphy.sims.blochsim``, ``mrphy.sims.freeprec``
    and ``mrphy.slowsims.blochsim_1step`` (the call targets of ``mrphy.mobjs``,
    ``mobjs.py:173,446,588``) with the HIP-backed functions.  ``mobjs`` looks them up as module
    attributes at call time, so ``SpinArray.applypulse`` etc. need no change.

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
    beffective.LAZY_DEFAULT = bool(lazy_beff)
    mrphy.beffective.rfgr2beff = beffective.rfgr2beff
    mrphy.sims.blochsim = sims.blochsim
    mrphy.slowsims.blochsim_1step = slowsims.blochsim_1step
    mrphy.sims.freeprec = sims.freeprec          # mobjs.SpinArray.freeprec (mobjs.py:588)
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
    beffective.LAZY_DEFAULT = False
    return mrphy
