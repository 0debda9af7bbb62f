"""Build the development variant of the library: ``tools/libmrphy_hip_dev.so`` = the product
sources compiled with ``-DMRPHY_DEV_KNOBS`` (environment knobs ``MRPHY_{K0,FWD,BWD}_VARIANT``,
``MRPHY_XCD_SWEEP`` that select alternative builds / block orders, and the per-workgroup time
stamps of ``mrphy_dev_set_stamps``).  The shipped ``mrphy.py_amd/libmrphy_hip.so`` has none of it.

    python tools/build_dev.py            # build if stale
    import tools.devlib; devlib.use()    # make mrphy_amd load the dev library (before first use)
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
# MRPHY_DEV_TAG / MRPHY_DEV_FLAGS: a second dev library with extra -D flags, e.g. an experiment build to A/B against
TAG = os.environ.get('MRPHY_DEV_TAG', '')
FLAGS = os.environ.get('MRPHY_DEV_FLAGS', '').split()
OUT = os.path.join(ROOT, 'tools', f'libmrphy_hip_dev{"_" + TAG if TAG else ""}.so')


def build(force=False):
    import mrphy_amd  # noqa: F401
    from mrphy_amd import _lib
    # MRPHY_DEV_UNIT_FLAGS="tu_x.hip=-flag -flag;tu_y.hip=..." : extra flags for single units of this dev library
    for item in filter(None, os.environ.get('MRPHY_DEV_UNIT_FLAGS', '').split(';')):
        u, f = item.split('=', 1)
        _lib.UNIT_FLAGS[u.strip()] = list(_lib.UNIT_FLAGS.get(u.strip(), ())) + f.split()
    st = _lib.build_library(OUT, os.path.join(ROOT, 'tools', 'build_dev' + ('_' + TAG if TAG else '')),
                            extra=['-DMRPHY_DEV_KNOBS'] + FLAGS, force=force)
    if st['compiled']:
        print(f"dev build: {st['compiled']} units in {st['seconds']} s", flush=True)
    return OUT


def use():
    r"""Point ``mrphy_amd`` at the dev library (call before the first kernel launch)."""
    import ctypes
    import mrphy_amd
    from mrphy_amd import _lib
    path = OUT if os.path.exists(OUT) else build()     # (a snapshot copy does not keep mtimes in order)
    _lib.library_path = lambda: path
    _lib._lib = None
    lib = _lib.require_library()
    lib.mrphy_dev_set_stamps.restype = ctypes.c_int
    lib.mrphy_dev_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
