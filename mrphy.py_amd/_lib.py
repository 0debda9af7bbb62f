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

ABI_VERSION = 5      # MRPHY_ABI_VERSION of include/mrphy_hip.h

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
    'mrphy_rfgr2beff_st': (_int, [_int, _vp, _i64, _vp, _i64, _vp] + _BC + _BC + [_vp, _vp]
                           + [_i64] * 4 + [_int, _vp]),
    'mrphy_rfgr2beff_bwd_workspace': (_sz, [_int] + [_i64] * 4),
    'mrphy_rfgr2beff_bwd': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i64] * 4 + [_vp]),
    'mrphy_blochsim_hist_bytes': (_sz, [_int] + [_i64] * 3),
    'mrphy_blochsim_fwd': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3
                           + [_vp]),
    'mrphy_blochsim_bwd': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp] + [_i64] * 3
                           + [_vp]),
    'mrphy_blochsim_bwd_consts': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp, _vp] + [_i64] * 3
                                  + [_vp]),
    'mrphy_blochsim_hist_part_bytes': (_sz, [_int] + [_i64] * 4),
    'mrphy_blochsim_fwd_parts': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp, _i64, _int] + [_i64] * 3
                                 + [_vp]),
    'mrphy_blochsim_bwd_parts': (_int, [_int, _vp, _i64, _int, _vp] + _BC * 3 + [_vp, _vp, _vp, _vp] + [_i64] * 3
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
    'mrphy_freeprec_bwd_consts': (_int, [_int, _vp, _vp, _vp, _i64] + _BC * 3 + [_vp, _i64, _i64, _vp]),
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
    'mrphy_beff2ab_bwd_consts': (_int, [_int, _vp, _vp] + _BC * 3 + [_vp, _vp, _vp, _vp] + [_i64] * 3 + [_vp]),
    'mrphy_blochsim_ab': (_int, [_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    'mrphy_blochsim_ab_bwd': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
}


def library_path() -> str:
    return os.path.join(_HERE, _LIBNAME)


# dtype codes as bits of -DMRPHY_DT_MASK: the launcher templates of a unit are instantiated for these codes
_ALL, _F32, _F64, _C64, _P, _PC64 = 0x1f, 1 << F32, 1 << F64, 1 << F32_C64, 1 << F32P, 1 << F32P_C64

# The translation units of the library: (source, dtype mask or None).  One hipcc job each, run in parallel;
# sorted by compile time, longest first (seconds on the 8-core build container, see build()).
UNITS = [(f, m) for f, masks in (
    ('tu_fused_mc_bwd.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_fused_fwd.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_fused_fwd1.hip', (_F32, _C64, _P, _PC64)),
    ('tu_fused_bwd.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_blochsim_bwd.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_blochsim_fwd.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_beff2ab.hip', (_F32, _F64, _C64, _P, _PC64)),
    ('tu_rfgr2beff_fwd.hip', (_F32, _F64)),
    ('tu_rfgr2beff_bwd.hip', (_F32, _F64)),
    ('tu_aux.hip', (None,)),
    ('abi.hip', (None,)),
) for m in masks]


def hipcc_flags() -> list:
    r"""Compile flags common to every unit."""
    return ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC',
            # fast-honor-pragmas, NOT fast: plain `fast` ignores `#pragma clang fp contract(off)`
            # (clang documents this), and then `p = s*E; q = p - off` is fused behind our back --
            # which silently breaks the error-free transformations of the precise step and the
            # "two roundings on z" the reference has (sims.py:77)
            '-ffp-contract=fast-honor-pragmas',
            # packed fp32 (v_pk_*_f32) has no throughput advantage on CDNA4 and the SLP vectoriser
            # pays for it with v_pk_mov/negate shuffles and hazard s_nops in the step loop
            '-fno-slp-vectorize',
            '-I', os.path.join(_HERE, os.pardir, 'include')]


def _hipcc() -> str:
    return os.environ.get('HIPCC') or os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'bin', 'hipcc')


def unit_object(objdir: str, src: str, mask) -> str:
    return os.path.join(objdir, os.path.splitext(src)[0] + ('' if mask is None else f'_dt{mask:02x}') + '.o')


# flags of single units (on top of hipcc_flags()): keyed by source name, or by (source name, dtype mask).
# The ILP-first scheduling strategy for the one-coil float builds of K2: 2-3 % faster, the checkpoint-writing build
# 15 % (64^3 x 2048: 0.81 -> 0.69 ms; tools/ab_libs_valu.py, profiles/r04_sched_ilp_ab.txt, r04_k2_ilp_ab.json).
# The multi-coil and fp64 builds lose with it (registers), K2b is indifferent: they keep the default.
_ILP = ('-mllvm', '-amdgpu-sched-strategy=max-ilp')
UNIT_FLAGS = {'tu_fused_fwd1.hip': _ILP}


def unit_command(src: str, mask, obj: str, extra=()) -> list:
    r"""The exact compile line of one unit."""
    return [_hipcc(), '-c'] + hipcc_flags() + list(UNIT_FLAGS.get(src, ())) + list(UNIT_FLAGS.get((src, mask), ())) + \
           list(extra) + \
           ([] if mask is None else [f'-DMRPHY_DT_MASK=0x{mask:02x}']) + \
           ['-MD', '-MF', obj + '.d', os.path.join(_CSRC, src), '-o', obj]


def link_command(objs, out: str) -> list:
    return [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + list(objs) + ['-o', out]


def _stamp(obj: str, tag: str):
    r"""What an up-to-date object was made from: its command line and the SHA-1 of every file of this
    repository it was compiled from (the -MD dependency list; ROCm's own headers are trusted).  Content,
    not mtimes: a snapshot copy of the tree (gpurun) does not keep mtimes in order."""
    import hashlib
    try:
        txt = open(obj + '.d').read().replace('\\\n', ' ')
    except OSError:
        return None
    root = os.path.realpath(os.path.join(_HERE, os.pardir))
    pkg = os.path.basename(_HERE)
    # The .d file holds the absolute paths of the tree the object was compiled in.  A copy of the tree under another
    # root (gpurun's snapshot, a user's checkout moved) must find the same files: a dependency is identified by its
    # path from the repository root DOWN -- `<pkg>/csrc/...` or `include/...` -- wherever that root was.
    rels = set()
    for d in (txt.split(':', 1)[1].split() if ':' in txt else ()):
        parts = os.path.normpath(d).split(os.sep)          # `<pkg>/../include/x.h` (the -I path) -> `include/x.h`
        if pkg in parts:
            rels.add(os.sep.join(parts[len(parts) - 1 - parts[::-1].index(pkg):]))
        elif 'include' in parts:
            cand = os.sep.join(parts[len(parts) - 1 - parts[::-1].index('include'):])
            if os.path.exists(os.path.join(root, cand)):    # ours, not one of ROCm's (those are trusted)
                rels.add(cand)
    lines = [tag]
    for r in sorted(rels):
        try:
            lines.append(r + ' ' + hashlib.sha1(open(os.path.join(root, r), 'rb').read()).hexdigest())
        except OSError:
            return None
    return '\n'.join(lines) if len(lines) > 1 else None


def kernel_count(path: str) -> int:
    r"""Kernels in the library's gfx950 code objects (kernel descriptor symbols ``<name>.kd``)."""
    import re
    with open(path, 'rb') as f:
        return len(set(re.findall(rb'[\x20-\x7e]{4,}\.kd\x00', f.read())))


def build_library(out: str, objdir: str, extra=(), force: bool = False, verbose: bool = False,
                  jobs: int = None) -> dict:
    r"""Compile the stale units in parallel (one hipcc process each) and link ``out``.  An object is stale if
    it is missing or its stamp (command line + content hashes of its sources, ``_stamp``) no longer matches.  Returns
    ``{'compiled': n, 'seconds': wall, 'units': {object name: seconds}}``."""
    import concurrent.futures
    import time
    os.makedirs(objdir, exist_ok=True)
    todo, objs = [], []
    for src, mask in UNITS:
        obj = unit_object(objdir, src, mask)
        cmd = unit_command(src, mask, obj, extra)
        objs.append(obj)
        tag = ' '.join(os.path.relpath(c, _HERE) if os.path.isabs(c) else c for c in cmd[1:])
        try:
            fresh = os.path.exists(obj) and open(obj + '.stamp').read() == _stamp(obj, tag)
        except OSError:
            fresh = False
        if force or not fresh:
            todo.append((obj, cmd, tag))
    t0 = time.time()
    times = {}

    def run(job):
        obj, cmd, tag = job
        t = time.time()
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"mrphy_amd: hipcc failed on {os.path.basename(obj)}\n{' '.join(cmd)}\n{r.stderr[-1800:]}")
        with open(obj + '.stamp', 'w') as f:
            f.write(_stamp(obj, tag) or '')
        times[os.path.basename(obj)] = round(time.time() - t, 1)

    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or max(1, os.cpu_count() or 1)) as ex:
            list(ex.map(run, todo))
    # The link has a stamp of its own: the objects' stamps, in link order.  A link that failed or was killed after the
    # objects were stamped, or a unit dropped from UNITS, leaves it stale although no object is.
    import hashlib
    link_tag = hashlib.sha1('\n'.join(
        [' '.join(os.path.relpath(c, _HERE) if os.path.isabs(c) else c for c in link_command(objs, out)[1:])]
        + [open(o + '.stamp').read() for o in objs]).encode()).hexdigest()
    try:
        linked = os.path.exists(out) and open(out + '.stamp').read() == link_tag
    except OSError:
        linked = False
    if todo or not linked:
        cmd = link_command(objs, out + '.tmp')
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(out + '.tmp', out)
        with open(out + '.stamp', 'w') as f:
            f.write(link_tag)
    return {'compiled': len(todo), 'linked': bool(todo or not linked), 'seconds': round(time.time() - t0, 1),
            'units': times}


def _objdir() -> str:
    return os.path.join(_HERE, 'build')


# ---------------------------------------------------------------------------------------------
# libmrphy_comm.so (include/mrphy_comm.h): the two collectives of the spin-sharded simulation over RCCL, for a
# consumer of the C ABI that has no torch.distributed.  Host code only; one translation unit; links librccl.so.1
# (in a process that has loaded PyTorch-ROCm that soname is already PyTorch's own copy).
# ---------------------------------------------------------------------------------------------
_COMM_LIBNAME = 'libmrphy_comm.so'
_comm = None
COMM_ABI_VERSION = 1     # MRPHY_COMM_ABI_VERSION of include/mrphy_comm.h
COMM_PROTOTYPES = {
    'mrphy_comm_abi_version': (_int, []),
    'mrphy_comm_error_string': (_c.c_char_p, [_int]),
    'mrphy_comm_unique_id': (_int, [_vp]),
    'mrphy_comm_init': (_int, [_vp, _int, _int, _c.POINTER(_vp)]),
    'mrphy_comm_destroy': (_int, [_vp]),
    'mrphy_comm_allgather_spins': (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    'mrphy_comm_allreduce_pulse_grads': (_int, [_vp, _vp, _i64, _int, _vp]),
}


def comm_library_path() -> str:
    return os.path.join(_HERE, _COMM_LIBNAME)


def comm_command(out: str) -> list:
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    return [_hipcc(), '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'c++', '-D__HIP_PLATFORM_AMD__',
            '-I', os.path.join(rocm, 'include'), '-I', os.path.join(_HERE, os.pardir, 'include'),
            os.path.join(_CSRC, 'comm.cpp'), '-L', os.path.join(rocm, 'lib'), '-lrccl', '-lamdhip64', '-o', out]


def build_comm(force: bool = False, verbose: bool = False) -> str:
    r"""Compile ``libmrphy_comm.so`` if it is missing or its stamp (command line + content of its two sources) is stale."""
    import hashlib
    out = comm_library_path()
    srcs = [os.path.join(_CSRC, 'comm.cpp'), os.path.join(_HERE, os.pardir, 'include', 'mrphy_comm.h')]
    cmd = comm_command(out + '.tmp')
    tag = hashlib.sha1((' '.join(os.path.relpath(c, _HERE) if os.path.isabs(c) else c for c in cmd[1:]) + '\n'
                        + '\n'.join(hashlib.sha1(open(f, 'rb').read()).hexdigest() for f in srcs)).encode()).hexdigest()
    try:
        fresh = os.path.exists(out) and open(out + '.stamp').read() == tag
    except OSError:
        fresh = False
    if force or not fresh:
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"mrphy_amd: building {_COMM_LIBNAME} failed\n{' '.join(cmd)}\n{r.stderr[-1800:]}")
        os.replace(out + '.tmp', out)
        with open(out + '.stamp', 'w') as f:
            f.write(tag)
    return out


def require_comm_library():
    r"""Load ``libmrphy_comm.so`` or raise (it pulls in RCCL: loaded only when the C-ABI collectives are asked for)."""
    global _comm
    if _comm is not None:
        return _comm
    with _lock:
        if _comm is not None:
            return _comm
        path = comm_library_path()
        if not os.path.exists(path):
            raise ImportError(f"mrphy_amd: {path} is missing. Build it first: mrphy_amd.build()")
        lib = ctypes.CDLL(path)
        for name, (res, args) in COMM_PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mrphy_comm_abi_version() != COMM_ABI_VERSION:
            raise ImportError(f"mrphy_amd: {path} has ABI version {lib.mrphy_comm_abi_version()}, this package binds "
                              f"{COMM_ABI_VERSION} (include/mrphy_comm.h): rebuild with mrphy_amd.build(force=True)")
        _comm = lib
    return _comm


def check_comm(code: int, what: str):
    if code != 0:
        msg = require_comm_library().mrphy_comm_error_string(code)
        raise RuntimeError(f"mrphy_amd: {what} failed with code {code}: {msg.decode() if msg else '?'}")


def build(force: bool = False, verbose: bool = False) -> str:
    r"""Compile ``libmrphy_hip.so`` for gfx950 if it is missing or older than its sources: the units of
    ``UNITS`` in parallel, then one link.  hipcc cross-compiles without a GPU, so this runs in the build
    container as well.  ``build.last`` holds the statistics of the last call."""
    global _lib
    out = library_path()
    with _lock:
        build.last = build_library(out, _objdir(), force=force, verbose=verbose)
        if build.last['compiled']:
            _lib = None
        try:
            build_comm(force=force, verbose=verbose)
        except (RuntimeError, OSError) as e:
            # the kernels' library is the product; the RCCL helpers are an extra for ctypes-only multi-GPU consumers:
            # a box without RCCL's headers still gets the former (require_comm_library() then raises ImportError)
            import sys
            print(f'mrphy_amd: {_COMM_LIBNAME} not built ({str(e).splitlines()[0]})', file=sys.stderr)
    return out


build.last = None


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
