r"""ctypes binding of ``libmrphy_hip.so`` (C ABI: ``include/mrphy_hip.h``) and its builder.

The library is built IN-TREE next to this file (``hipcc --offload-arch=gfx950 -shared``) so
that it travels with the source tree.  Nothing here falls back to another implementation: a
missing library, a CPU tensor or a non-zero return code raises.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
_LIBNAME = 'libmrphy_hip.so'
_lock = threading.Lock()
_lib = None

ABI_VERSION = 2      # MRPHY_ABI_VERSION of include/mrphy_hip.h

# dtype codes of mrphy_hip.h
F32, F64, F32_C64, F32P, F32P_C64 = 0, 1, 2, 3, 4

_c = ctypes
_vp, _i64, _int, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t
_BC = [_vp, _i64, _i64]            # broadcastable per-spin constant: ptr, stride_n, stride_m

# name -> (restype, argtypes); MUST list every function declared in include/mrphy_hip.h
PROTOTYPES = {
    'mrphy_abi_version': (_int, []),
    'mrphy_error_string': (_c.c_char_p, [_int]),
    'mrphy_arch': (_c.c_char_p, []),
    'mrphy_debug_xcc_map': (_int, [_vp, _i64, _vp]),
    'mrphy_rfgr2beff': (_int, [_int, _vp, _i64, _vp, _i64, _vp] + _BC + _BC + [_vp, _vp]
                        + [_i64] * 4 + [_vp]),
    'mrphy_rfgr2beff_bwd_workspace': (_sz, [_int] + [_i64] * 4),
    'mrphy_rfgr2beff_bwd': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i64] * 4 + [_vp]),
    'mrphy_blochsim_hist_bytes': (_sz, [_int] + [_i64] * 3),
    'mrphy_blochsim_fwd': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3
                           + [_vp]),
    'mrphy_blochsim_bwd': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3
                           + [_vp]),
    'mrphy_blochsim_bwd_consts': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp, _vp] + [_i64] * 3
                                  + [_vp]),
    'mrphy_blochsim_1step': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp] + [_i64] * 2 + [_vp]),
    'mrphy_blochsim_rfgr_fwd': (_int, [_int, _vp, _vp, _i64, _vp, _i64, _vp] + _BC + _BC + [_vp]
                                + _BC * 3 + [_vp, _vp, _vp, _i64] + [_i64] * 4 + [_vp]),
    'mrphy_blochsim_rfgr_ck_every': (_i64, []),
    'mrphy_blochsim_rfgr_bwd_workspace': (_sz, [_int] + [_i64] * 3),
    'mrphy_blochsim_rfgr_bwd': (_int, [_int, _vp, _vp, _i64, _vp, _i64, _vp] + _BC + _BC + [_vp]
                                + _BC * 3 + [_vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i64] * 3 + [_vp]),
    'mrphy_blochsim_rfgr_mc_max_coils': (_i64, []),
    'mrphy_blochsim_rfgr_mc_bwd_workspace': (_sz, [_int] + [_i64] * 4),
    'mrphy_blochsim_rfgr_mc_bwd': (_int, [_int, _vp, _vp, _i64, _vp, _i64, _vp] + _BC + _BC + [_vp]
                                   + _BC * 3 + [_vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i64] * 4 + [_vp]),
    'mrphy_freeprec_fwd': (_int, [_int, _vp, _vp, _i64] + _BC * 3 + [_vp, _i64, _i64, _vp]),
    'mrphy_freeprec_bwd': (_int, [_int, _vp, _vp, _i64] + _BC * 3 + [_vp, _i64, _i64, _vp]),
    'mrphy_pulse_interp_linear': (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_pulse_interp_select': (_int, [_int, _int, _vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_beff2uphi': (_int, [_int, _vp] + _BC + [_vp, _vp, _i64, _i64, _vp]),
    'mrphy_uphirot': (_int, [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'mrphy_beff2uphi_bwd': (_int, [_int, _vp] + _BC + [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'mrphy_uphirot_bwd': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'mrphy_mask_extract': (_int, [_int, _vp, _vp, _vp] + [_i64] * 4 + [_vp]),
    'mrphy_mask_embed': (_int, [_int, _vp, _vp, _vp] + [_i64] * 4 + [_int, ctypes.c_uint64, _vp]),
    'mrphy_cube_loc': (_int, [_int, _vp, _vp, _vp, _vp] + [_i64] * 5 + [_vp]),
    'mrphy_beff2ab': (_int, [_int, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_beff2ab_hist_bytes': (_sz, [_int] + [_i64] * 3),
    'mrphy_beff2ab_save': (_int, [_int, _vp] + _BC * 3 + [_vp, _vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_beff2ab_bwd': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_blochsim_ab': (_int, [_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    'mrphy_blochsim_ab_bwd': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
}


def library_path() -> str:
    return os.path.join(_HERE, _LIBNAME)


def hipcc_command(out: str = None) -> list:
    r"""The exact compile line for the gfx950 shared library."""
    hipcc = os.environ.get('HIPCC') or os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'),
                                                    'bin', 'hipcc')
    return [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
            # fast-honor-pragmas, NOT fast: plain `fast` ignores `#pragma clang fp contract(off)`
            # (clang documents this), and then `p = s*E; q = p - off` is fused behind our back --
            # which silently breaks the error-free transformations of the precise step and the
            # "two roundings on z" the reference has (sims.py:77)
            '-ffp-contract=fast-honor-pragmas',
            # packed fp32 (v_pk_*_f32) has no throughput advantage on CDNA4 and the SLP vectoriser
            # pays for it with v_pk_mov/negate shuffles and hazard s_nops in the step loop
            '-fno-slp-vectorize',
            '-I', os.path.join(_HERE, os.pardir, 'include'),
            os.path.join(_CSRC, 'mrphy_hip.hip'), '-o', out or library_path()]


def _sources():
    return [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))
            if f.endswith(('.hip', '.hpp', '.h'))] + \
           [os.path.join(_HERE, os.pardir, 'include', 'mrphy_hip.h'),
            os.path.abspath(__file__)]          # the compile flags live in this file


def build(force: bool = False, verbose: bool = False) -> str:
    r"""Compile ``libmrphy_hip.so`` for gfx950 if it is missing or older than its sources.

    hipcc cross-compiles without a GPU, so this runs in the build container as well.
    """
    global _lib
    out = library_path()
    with _lock:
        stale = force or not os.path.exists(out)
        if not stale:
            mt = os.path.getmtime(out)
            stale = any(os.path.getmtime(s) > mt for s in _sources() if os.path.exists(s))
        if stale:
            cmd = hipcc_command(out + '.tmp')
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(out + '.tmp', out)
            _lib = None
    return out


def require_library():
    r"""Load the shared library or raise: there is no other implementation to fall back on."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"mrphy_amd: {path} is missing. Build it first: "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)       # AttributeError here = library/header out of sync
            fn.restype, fn.argtypes = res, args
        if lib.mrphy_abi_version() != ABI_VERSION:
            raise ImportError(f"mrphy_amd: {path} has ABI version {lib.mrphy_abi_version()}, this "
                              f"package binds version {ABI_VERSION} (include/mrphy_hip.h): stale "
                              "library -- rebuild with mrphy_amd.build(force=True)")
        _lib = lib
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = require_library().mrphy_error_string(code)
        raise RuntimeError(f"mrphy_amd: {what} failed with code {code}: "
                           f"{msg.decode() if msg else '?'}")
