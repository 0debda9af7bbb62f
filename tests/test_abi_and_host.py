r"""No-GPU checks: the C-ABI library builds/loads and exports exactly what include/*.h declares,
the ctypes prototypes agree with the header, the host-side argument plumbing does what the
reference's wrappers do, and the product path refuses to run without a device (no fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

import mrphy_amd
from mrphy_amd import _host, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mrphy_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mrphy_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = mrphy_amd.build()                      # hipcc cross-compiles for gfx950 without a GPU
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f'{n} declared in mrphy_hip.h but not exported'
    assert sorted(_lib.PROTOTYPES) == names, 'ctypes prototypes out of sync with the header'
    lib2 = mrphy_amd.require_library()
    assert lib2.mrphy_abi_version() == _lib.ABI_VERSION == 5 and lib2.mrphy_arch() == b'gfx950'
    assert lib2.mrphy_error_string(-1) == b'mrphy: invalid argument'


def test_comm_library_exports_every_declared_symbol_and_checks_arguments():
    r"""``include/mrphy_comm.h`` / ``libmrphy_comm.so`` (the RCCL helpers behind the C ABI): built in-tree next to
    the kernels' library, every declared function exported and bound, argument errors caught before RCCL is entered.  No
    collective runs here (no GPU): that is ``tests/test_bench_dist.py::test_c_abi_collectives_equal_torch_distributed``."""
    mrphy_amd.build()
    src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'mrphy_comm.h')).read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(mrphy_comm_[a-z0-9_]+)\s*\(', src)))
    assert len(names) == 7 and sorted(_lib.COMM_PROTOTYPES) == names, names
    raw = ctypes.CDLL(_lib.comm_library_path())
    for n in names:
        assert hasattr(raw, n), f'{n} declared in mrphy_comm.h but not exported'
    lib = _lib.require_comm_library()
    assert lib.mrphy_comm_abi_version() == _lib.COMM_ABI_VERSION == 1
    assert lib.mrphy_comm_error_string(0) == b'success' and b'invalid argument' in lib.mrphy_comm_error_string(-1)
    h = ctypes.c_void_p()
    assert lib.mrphy_comm_init(None, 1, 0, ctypes.byref(h)) == -1                      # no id
    assert lib.mrphy_comm_init(b'x' * 128, 0, 0, ctypes.byref(h)) == -1                # no ranks
    assert lib.mrphy_comm_init(b'x' * 128, 2, 2, ctypes.byref(h)) == -1                # rank out of range
    assert lib.mrphy_comm_allgather_spins(None, None, None, 4, 0, None) == -1          # no communicator
    assert lib.mrphy_comm_allreduce_pulse_grads(None, None, 4, 0, None) == -1
    assert lib.mrphy_comm_destroy(None) == 0
    # the kernels' library does not depend on RCCL; the comm library does
    deps = lambda f: subprocess.run(['readelf', '-d', f], capture_output=True, text=True).stdout  # noqa: E731
    assert 'librccl' not in deps(mrphy_amd.library_path()) and 'librccl.so.1' in deps(_lib.comm_library_path())
    # the Python route: dist.use_c_abi swaps the collectives of mrphy_amd.dist
    from mrphy_amd import dist as D
    assert D.use_c_abi(None) is None and D._C_COMM is None


def test_code_object_is_gfx950_only():
    # llvm-objdump --offloading writes one unbundled code object per unit NEXT TO ITS INPUT: give it a
    # symlink in a temporary directory, or 80 files land in the package directory (and travel with gpurun)
    import tempfile
    with tempfile.TemporaryDirectory(prefix='mrphy_co_') as d:
        link = os.path.join(d, os.path.basename(mrphy_amd.library_path()))
        os.symlink(os.path.abspath(mrphy_amd.library_path()), link)
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', link], cwd=d,
                             capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip('llvm-objdump --offloading unavailable')
    assert not [f for f in os.listdir(os.path.dirname(mrphy_amd.library_path())) if 'hipv4-' in f or '.host-' in f]
    archs = set(re.findall(r'gfx[0-9a-f]+', out.stdout))
    assert archs == {'gfx950'}, archs


def test_argument_errors_are_caught_on_the_host():
    lib = mrphy_amd.require_library()
    # unknown dtype / negative size / null pointers: rejected before any HIP call
    assert lib.mrphy_blochsim_fwd(9, None, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None,
                                  None, 1, 1, 1, None) == -1
    assert lib.mrphy_blochsim_fwd(0, None, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None,
                                  None, 1, -1, 1, None) == -1
    assert lib.mrphy_blochsim_fwd(0, None, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None,
                                  None, 1, 4, 8, None) == -1
    # empty problems are a no-op success
    assert lib.mrphy_blochsim_fwd(0, None, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None,
                                  None, 0, 0, 8, None) == 0
    # the history in parts (ABI 5): part count 0..8, a known layout, no null part; sizes by the query
    import ctypes
    two = (ctypes.c_void_p * 2)(None, None)
    fwd_parts = lambda parts, n, layout, nM=4: lib.mrphy_blochsim_fwd_parts(  # noqa: E731
        0, None, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None, parts, n, layout, 1, nM, 8, None)
    assert fwd_parts(None, 9, 0) == -1 and fwd_parts(None, -1, 0) == -1 and fwd_parts(None, 0, 2) == -1
    assert fwd_parts(two, 2, 0) == -1                     # null part pointers
    assert fwd_parts(None, 0, 0, nM=0) == 0               # empty problem: no-op success
    assert lib.mrphy_blochsim_bwd_parts(0, two, 2, 1, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None, None,
                                        None, 1, 4, 8, None) == -1
    assert lib.mrphy_blochsim_hist_bytes(0, 1, 1000, 64) == 16 * 64 * 192 * 4
    assert lib.mrphy_blochsim_hist_part_bytes(0, 1, 1000, 64, 1) == 16 * 64 * 192 * 4
    assert lib.mrphy_blochsim_hist_part_bytes(0, 1, 1000, 64, 3) == 6 * 64 * 192 * 4      # ceil(16 tiles / 3)
    assert lib.mrphy_blochsim_hist_part_bytes(1, 1, 1000, 64, 8) == 2 * 64 * 192 * 8
    assert lib.mrphy_blochsim_hist_part_bytes(0, 1, 1000, 64, 9) == 0
    assert lib.mrphy_rfgr2beff_bwd_workspace(0, 1, 1000, 64, 1) == 4 * 9 * 64 * 4
    # 2..32 coils: (3 + 2 nC) partial rows x nT per spin group (4 groups of <= 256 spins), rounded up to 256 B,
    # + per spin a packed coefficient row [b1r | b1i | loc, 0] of 2 MC + 4 words, MC = nC padded to 4/8/12/16/24/32
    up = lambda b: (b + 255) // 256 * 256  # noqa: E731
    assert lib.mrphy_rfgr2beff_bwd_workspace(0, 1, 1000, 64, 2) == up(4 * 7 * 64 * 4) + 1000 * (2 * 4 + 4) * 4
    assert lib.mrphy_rfgr2beff_bwd_workspace(0, 1, 1000, 64, 9) == up(4 * 21 * 64 * 4) + 1000 * (2 * 12 + 4) * 4
    assert lib.mrphy_rfgr2beff_bwd_workspace(1, 2, 1000, 64, 32) == up(4 * 2 * 67 * 64 * 8) + 2000 * (2 * 32 + 4) * 8
    # > 32 coils (round 4): the same pass over blocks of 32 coils -- all rows of partial sums + ONE block's packed rows
    assert lib.mrphy_rfgr2beff_bwd_workspace(0, 1, 1000, 64, 33) == up(4 * 69 * 64 * 4) + 1000 * (2 * 32 + 4) * 4


def test_no_cpu_fallback():
    M = torch.zeros(1, 4, 3)
    B = torch.zeros(1, 4, 8, 3)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        mrphy_amd.sims.blochsim(M, B)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        mrphy_amd.beffective.rfgr2beff(torch.zeros(1, 2, 8), torch.zeros(1, 3, 8), torch.zeros(1, 4, 3))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        mrphy_amd.slowsims.blochsim_1step(M, M, M, *(torch.ones(()),) * 4)
    # and the reference's shape asserts come first, as in sims.py:305,311
    with pytest.raises(AssertionError):
        mrphy_amd.sims.blochsim(M, torch.zeros(1, 5, 8, 3))
    with pytest.raises(AssertionError):
        mrphy_amd.sims.blochsim(M, B, T1=torch.ones(()))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'mrphy.py_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert 'bloch_oracle' not in src, f
            assert not re.search(r'^\s*(import|from)\s+\S*oracle', src, flags=re.M), f
            assert "'oracle'" not in src and '"oracle"' not in src, f   # no sys.path games either


def test_fused_adjoints_pin_their_segment_length():
    r"""`k_bloch_rfgr_bwd_mc` takes its step from the lane (`lane >> 2`) while its LDS arrays and
    workspace rows are sized from `geom.hpp`'s SEG -- a SEG = 8 dev build read past `raw[]` and stored past the
    segment (GPU fault).  The kernels now refuse to compile with any other SEG."""
    csrc = os.path.join(ROOT, 'mrphy.py_amd', 'csrc')
    mc = open(os.path.join(csrc, 'k_fused_mc_bwd.hpp')).read()
    common = open(os.path.join(csrc, 'k_fused_bwd_common.hpp')).read()
    geom = open(os.path.join(csrc, 'geom.hpp')).read()
    assert re.search(r'static_assert\(SEG \* 4 == WAVE && SEG == 16', mc)
    assert mc.index('static_assert(SEG * 4 == WAVE') < mc.index('st_w = lane >> 2')   # ahead of the first use
    assert re.search(r'static_assert\(SEG == 16', common) and re.search(r'static_assert\(SEG % 4 == 0', common)
    assert 'constexpr int SEG = 16;' in geom and 'NOT generic' in geom


def test_signatures_match_the_reference():
    import inspect
    sig = inspect.signature(mrphy_amd.sims.blochsim)
    assert list(sig.parameters)[:6] == ['Mi', 'Beff', 'T1', 'T2', 'γ', 'dt']          # + the `workspace=` extension
    assert list(sig.parameters)[6:] == ['workspace'] and sig.parameters['workspace'].default is None
    assert all(sig.parameters[k].kind is inspect.Parameter.KEYWORD_ONLY for k in ('T1', 'T2', 'γ', 'dt'))
    assert sig.parameters['γ'].default is mrphy_amd.γH and sig.parameters['dt'].default is mrphy_amd.dt0
    sig = inspect.signature(mrphy_amd.beffective.rfgr2beff)
    assert list(sig.parameters)[:6] == ['rf', 'gr', 'loc', 'Δf', 'b1Map', 'γ']
    sig = inspect.signature(mrphy_amd.slowsims.blochsim_1step)
    assert list(sig.parameters) == ['M', 'M1', 'b', 'E1', 'E1_1', 'E2', 'γ2πdt']
    for c in (mrphy_amd.γH, mrphy_amd.dt0, mrphy_amd.T1G, mrphy_amd.T2G):
        assert c.dtype == torch.float64 and c.ndim == 0            # mrphy/__init__.py:58-65
    assert float(mrphy_amd.γH) == 4257.6 and float(mrphy_amd.dt0) == 4e-6


def test_bcast_descriptors():
    N, Nd = 2, (5,)
    f = torch.float32
    mk = lambda x: _host.Bcast(x, N, Nd, f, torch.device('cpu'))  # noqa: E731
    assert (mk(torch.tensor(3.)).sn, mk(torch.tensor(3.)).sm) == (0, 0)
    assert (mk(torch.ones(1, 1)).sn, mk(torch.ones(1, 1)).sm) == (0, 0)
    b = mk(torch.arange(10.).reshape(2, 5))
    assert (b.sn, b.sm) == (5, 1)
    b = mk(torch.arange(2.).reshape(2, 1).expand(2, 5))         # stride-0 view, as mobjs keeps T1_
    assert (b.sn, b.sm) == (1, 0)
    b = mk(torch.ones(1, 1).expand(2, 5))
    assert (b.sn, b.sm) == (0, 0)
    b = mk(torch.arange(5.).reshape(1, 5, 1, 1))                # padded to the rank of Beff
    assert (b.sn, b.sm) == (0, 1)
    b = mk(torch.arange(2.).reshape(2, 1, 1, 1))                # dt (N,) padded (sims.py:309)
    assert (b.sn, b.sm) == (1, 0)
    with pytest.raises(AssertionError):
        mk(torch.ones(3, 5))
    # general *Nd is flattened
    b = _host.Bcast(torch.arange(12.).reshape(1, 3, 4), 2, (3, 4), f, torch.device('cpu'))
    assert (b.sn, b.sm) == (0, 1) and b.t.shape == (1, 12)
    # constant dtype follows torch promotion of the reference's expressions
    assert _host.precision.get() == 'precise'                    # the default step arithmetic
    assert _host.dtype_code(torch.float32, torch.float32) == _lib.F32P == 3
    assert _host.dtype_code(torch.float32, torch.float64) == _lib.F32P_C64 == 4
    assert _host.dtype_code(torch.float64, torch.float32) == _lib.F64
    with _host.precision('fast'):
        assert _host.dtype_code(torch.float32, torch.float32) == _lib.F32 == 0
        assert _host.dtype_code(torch.float32, torch.float64) == _lib.F32_C64 == 2
        assert _host.dtype_code(torch.float64, torch.float64) == _lib.F64
    assert _host.precision.get() == 'precise'
    g, E1, E2, E1_1 = mrphy_amd.sims._gamma_dt_constants(
        torch.ones(1, 1, 1, 1), torch.ones(1, 1, 1, 1), mrphy_amd.γH.reshape(1, 1, 1, 1),
        torch.tensor([4e-6]).reshape(1, 1, 1, 1))
    assert g.dtype == torch.float64 and E1.dtype == torch.float32   # fp64 γH default meets fp32 dt


def test_shard_bounds_cover_all_spins():
    from mrphy_amd.dist import shard_bounds
    for nM in (1, 7, 64, 1000, 128 ** 3):
        for ws in (1, 2, 3, 8):
            b = [shard_bounds(nM, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == nM
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.skipif(not os.path.isdir('/root/reference/mrphy'), reason='reference only in the build container')
def test_install_routes_the_reference_object_layer():
    r"""With the reference importable (build container only): after install(), the function targets of mobjs are
    routers (SURVEY 8b "Fallback"): a call whose tensors all live on the CPU is handed to the
    reference's OWN saved function -- so a CPU ``cube.applypulse(p)`` after install() equals the uninstalled result
    bit for bit -- and a call with any tensor elsewhere goes to the HIP path and never reaches the saved reference
    callable (checked here with tensors on the 'meta' device and the saved callables replaced by tripwires; on the
    GPU box by test_install_router_sends_device_tensors_to_the_kernels)."""
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
import torch, mrphy, mrphy_amd
from mrphy import mobjs
cube, p = mobjs.Examples.spincube(), mobjs.Examples.pulse()
want = cube.applypulse(p)                        # the reference, before install()
want_fp = cube.freeprec(torch.tensor(1e-3))
want_b = cube.pulse2beff(p) if False else mrphy.beffective.rfgr2beff(p.rf, p.gr, cube.loc_)
mrphy_amd.install(mrphy)
for name, fn, hip in (('blochsim', mrphy.sims.blochsim, mrphy_amd.sims.blochsim),
                      ('rfgr2beff', mrphy.beffective.rfgr2beff, mrphy_amd.beffective.rfgr2beff),
                      ('blochsim_1step', mrphy.slowsims.blochsim_1step, mrphy_amd.slowsims.blochsim_1step),
                      ('freeprec', mrphy.sims.freeprec, mrphy_amd.sims.freeprec),
                      ('beff2ab', mrphy.beffective.beff2ab, mrphy_amd.beffective.beff2ab),
                      ('blochsim_ab', mrphy.slowsims.blochsim_ab, mrphy_amd.slowsims.blochsim_ab)):
    assert fn is mrphy_amd._routed[name] and fn.hip is hip and fn.__wrapped__ is hip, name
    assert mrphy_amd._saved[name].__module__.startswith('mrphy.'), name
# object-layer glue (SURVEY 8f-3): objects on the CPU keep the reference's own gather/scatter
assert mobjs.SpinArray.extract is mrphy_amd._spinarray_extract
assert mobjs.SpinCube._update_loc_ is mrphy_amd._spincube_update_loc_
assert mobjs.Pulse.interpT is mrphy_amd._pulse_interpT
assert mobjs.SpinArray.applypulse is mrphy_amd._spinarray_applypulse
p2 = p.interpT(p.dt / 2)
assert p2.rf.shape[2] == 2 * p.rf.shape[2] and 'interpT' in p2.desc and p2.device == p.device
assert torch.equal(cube.extract(cube.embed(cube.M_)), cube.M_)
# CPU objects after install(): the user's own reference, bit for bit
assert torch.equal(cube.applypulse(p), want)
assert torch.equal(cube.freeprec(torch.tensor(1e-3)), want_fp)
assert torch.equal(mrphy.beffective.rfgr2beff(p.rf, p.gr, cube.loc_), want_b)
# ... while mrphy_amd's own functions keep refusing CPU tensors (no CPU path in the product)
try:
    mrphy_amd.sims.blochsim(cube.M_, want_b)
except RuntimeError as e:
    assert 'no CPU fallback' in str(e), e
else:
    raise SystemExit('mrphy_amd.sims.blochsim computed on the CPU')
# a tensor that is NOT on the CPU never reaches the saved reference callables
def tripwire(*a, **k):
    raise SystemExit('a non-CPU call reached the saved reference function')
real = dict(mrphy_amd._saved)
for k in ('rfgr2beff', 'blochsim', 'blochsim_1step', 'freeprec', 'beff2ab', 'blochsim_ab'):
    mrphy_amd._saved[k] = tripwire
meta = lambda *s: torch.empty(*s, device='meta')
calls = ((mrphy.sims.blochsim, (meta(1, 9, 3), meta(1, 9, 8, 3)), {}),
         (mrphy.sims.blochsim, (cube.M_, meta(1, 9, 8, 3)), {}),                 # mixed: one device tensor is enough
         (mrphy.beffective.rfgr2beff, (meta(1, 2, 8), meta(1, 3, 8), meta(1, 9, 3)), {}),
         (mrphy.sims.freeprec, (meta(1, 9, 3), meta(1)), {}),
         (mrphy.slowsims.blochsim_1step, (meta(1, 9, 3), meta(1, 9, 3), meta(1, 9, 3)) + (meta(1, 1),) * 4, {}),
         (mrphy.beffective.beff2ab, (meta(1, 9, 8, 3),), {}),
         (mrphy.slowsims.blochsim_ab, (meta(1, 9, 3), meta(1, 9, 3, 3), meta(1, 9, 3)), {}))
for fn, a, k in calls:
    try:
        fn(*a, **k)
    except SystemExit:
        raise
    except Exception as e:                      # the HIP host layer's own refusal of a non-ROCm device
        assert 'mrphy_amd' in str(e) or isinstance(e, (AssertionError, NotImplementedError)), (fn, e)
    else:
        raise SystemExit(f'{fn} accepted a meta tensor')
mrphy_amd._saved.update(real)
mrphy_amd.uninstall(mrphy)
assert not mrphy_amd._routed and not mrphy_amd._saved
assert mrphy.sims.freeprec.__module__ == 'mrphy.sims'
assert mrphy.sims.blochsim.__module__ == 'mrphy.sims' and not hasattr(mrphy.sims.blochsim, 'hip')
assert mobjs.SpinArray.extract.__module__ == 'mrphy.mobjs'
assert mobjs.Pulse.interpT.__module__ == 'mrphy.mobjs'
assert mobjs.SpinArray.applypulse.__module__ == 'mrphy.mobjs'
mrphy_amd.install(mrphy, fuse_applypulse=False)
assert mobjs.SpinArray.applypulse.__module__ == 'mrphy.mobjs' and mrphy.sims.blochsim.hip is mrphy_amd.sims.blochsim
assert torch.equal(cube.applypulse(p), want)
mrphy_amd.uninstall(mrphy)
assert mobjs.SpinArray.applypulse.__module__ == 'mrphy.mobjs'
assert torch.equal(cube.applypulse(p), want)          # the reference again
print('routed')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and 'routed' in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_interp_grid_matches_reference_quirks():
    r"""Host side of the on-device interpT: sample counts and weights as Pulse.interpT forms them
    (mobjs.py:211-212), checked against the reference's known answer (tests/test_mobjs.py:160-195)
    and the golden sample counts (incl. the 255-sample float-floor quirk)."""
    import numpy as np
    from mrphy_amd.interp import interp_grid
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import golden
    I = golden('interp_f32')
    f32 = lambda v: torch.tensor(v, dtype=torch.float32).item()  # noqa: E731
    assert interp_grid(1024, f32(8e-6), f32(4e-6))[3] == I['rf'].shape[2] == 2048
    assert interp_grid(512, f32(4e-6), 8e-6)[3] == int(I['quirk_nT']) == 255
    # known answer: nT = 11 ramps, dt -> 5 dt (fp64)
    nT, dt = 11, 4e-6
    rf = 0.1 * np.concatenate([np.linspace([[0.]], 1., num=nT, axis=2), np.linspace([[1]], 0., num=nT, axis=2)], 1)
    lo, w, dx, n = interp_grid(nT, dt, 5 * dt)
    ext = np.concatenate([np.zeros_like(rf[:, :, :1]), rf], axis=2)
    out = (ext[:, :, lo + 1] - ext[:, :, lo]) / dx * w + ext[:, :, lo]
    assert n == 2 and np.allclose(out, np.array([[[0.04, 0.09], [0.06, 0.01]]]), atol=1e-9)


def test_argument_errors_of_the_newer_entry_points():
    r"""§8f rows and the parallel-transmit adjoint: bad arguments are rejected on the host
    (MRPHY_EINVAL = -1, MRPHY_ENOSPC = -3), empty problems succeed, workspace sizes are as
    documented -- no HIP call is made in any of these."""
    lib = mrphy_amd.require_library()
    EINVAL, ENOSPC = -1, -3
    # mask gather/scatter: element size, nM > nV, negative sizes; empty is fine
    assert lib.mrphy_mask_extract(2, None, None, None, 1, 8, 4, 3, None) == EINVAL
    assert lib.mrphy_mask_extract(4, None, None, None, 1, 4, 8, 3, None) == EINVAL
    assert lib.mrphy_mask_extract(4, None, None, None, 1, 8, 4, 3, None) == EINVAL      # null pointers
    assert lib.mrphy_mask_extract(4, None, None, None, 1, 8, 0, 3, None) == 0
    assert lib.mrphy_mask_embed(8, None, None, None, 1, 8, 4, 3, 7, 0, None) == EINVAL   # fill flag
    assert lib.mrphy_mask_embed(8, None, None, None, 0, 8, 4, 3, 1, 0, None) == 0
    assert lib.mrphy_cube_loc(0, None, None, None, None, 1, 4, 0, 2, 2, None) == EINVAL  # nx < 1
    assert lib.mrphy_cube_loc(0, None, None, None, None, 1, 9, 2, 2, 2, None) == EINVAL  # nM > nV
    assert lib.mrphy_cube_loc(0, None, None, None, None, 1, 0, 2, 2, 2, None) == 0
    # A/B propagation
    assert lib.mrphy_beff2ab(7, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None, None,
                             1, 1, 1, None) == EINVAL
    assert lib.mrphy_beff2ab(0, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None, None,
                             1, 4, 8, None) == EINVAL                                     # nulls
    assert lib.mrphy_beff2ab(0, None, None, 0, 0, None, 0, 0, None, 0, 0, None, None, None,
                             1, 0, 8, None) == 0
    assert lib.mrphy_blochsim_ab(2, None, None, None, None, 4, None) == EINVAL             # no F32_C64
    assert lib.mrphy_blochsim_ab(0, None, None, None, None, 0, None) == 0
    assert lib.mrphy_blochsim_ab_bwd(0, None, None, None, None, None, 4, None) == 0        # nothing wanted
    # parallel-transmit fused adjoint
    assert lib.mrphy_blochsim_rfgr_mc_max_coils() == 8
    ck = lib.mrphy_blochsim_rfgr_ck_every()
    assert lib.mrphy_blochsim_rfgr_mc_bwd_workspace(0, 1, 64 * 5000, 2 * ck, 4) == \
        2048 * 1 * (3 + 2 * 4) * 2 * ck * 4                    # persistent waves x rows x nT x 4 B
    assert lib.mrphy_blochsim_rfgr_mc_bwd_workspace(1, 1, 100, ck, 2) == 2 * 7 * ck * 8
    args = [None, None, 0, None, 0, None] + [None, 0, 0] * 2 + [None] + [None, 0, 0] * 3 + [None]
    tail = [None, None, None, None, None, 0]
    assert lib.mrphy_blochsim_rfgr_mc_bwd(0, *args, *tail, 1, 64, ck, 9, None) == EINVAL    # 9 coils
    assert lib.mrphy_blochsim_rfgr_mc_bwd(0, *args, *tail, 1, 64, ck + 1, 4, None) == EINVAL  # nT % 16
    assert lib.mrphy_blochsim_rfgr_mc_bwd(0, *args, *tail, 1, 0, ck, 4, None) == 0


def test_mask_index_is_keyed_by_identity(monkeypatch):
    r"""install()'s per-mask index cache: a second lookup of the same mask tensor must hit the cache
    without comparing tensors with == (weakref.WeakKeyDictionary did: Tensor.__eq__ is elementwise
    and bool() of it raises), equal-valued but distinct masks get their own entry, and entries die
    with their mask.  MaskIndex itself needs the GPU; a stand-in records what it was built from."""
    import gc
    built = []

    class FakeIndex:
        def __init__(self, mask):
            built.append(id(mask))

    monkeypatch.setattr(mrphy_amd.masks, 'MaskIndex', FakeIndex)
    monkeypatch.setattr(mrphy_amd, '_mask_index', None)
    m = torch.ones((1, 3, 4), dtype=torch.bool)
    a = mrphy_amd._index_of(m)
    assert mrphy_amd._index_of(m) is a and mrphy_amd._index_of(m) is a and len(built) == 1
    m2 = m.clone()
    assert mrphy_amd._index_of(m2) is not a and len(built) == 2
    assert len(mrphy_amd._mask_index) == 2
    del m2
    gc.collect()
    assert len(mrphy_amd._mask_index) == 1


def test_precision_knob():
    import subprocess
    assert mrphy_amd.precision.get() == 'precise'
    with mrphy_amd.precision('fast'):
        assert mrphy_amd.precision.get() == 'fast'
        with mrphy_amd.precision('precise'):
            assert mrphy_amd.precision.get() == 'precise'
        assert mrphy_amd.precision.get() == 'fast'
    assert mrphy_amd.precision.get() == 'precise'
    code = "import sys; sys.path.insert(0, %r); import mrphy_amd; print(mrphy_amd.precision.get())" % ROOT
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MRPHY_PRECISION='fast'),
                         capture_output=True, text=True)
    assert out.stdout.strip() == 'fast', out.stderr[-500:]
    # the header names the two precise codes the host sends
    hdr = open(os.path.join(ROOT, 'include', 'mrphy_hip.h')).read()
    assert '#define MRPHY_F32P     3' in hdr and '#define MRPHY_F32P_C64 4' in hdr


@pytest.mark.parametrize('kind', ['nearest', 'nearest-up', 'previous', 'next', 'zero'])
def test_interp_select_index_is_scipys(kind):
    r"""Host side of the one-tap interpT kinds: selecting by ``interp_select``'s index reproduces
    ``scipy.interpolate.interp1d(kind=kind)`` -- the call ``Pulse.interpT`` makes, ``mobjs.py:214-215``
    -- BIT FOR BIT on random waveforms, on the grids the reference builds (``mobjs.py:209-212``):
    up- and down-sampling, non-integer ratios, float32-rounded dwell times, the 255-sample floor
    quirk.  The index is in range and non-decreasing: what the kernel relies on."""
    import numpy as np
    from scipy import interpolate
    from mrphy_amd.interp import interp_select, interp_grid, SELECT_KINDS
    assert kind in SELECT_KINDS
    f32 = lambda v: float(np.float32(v))  # noqa: E731
    rng = np.random.default_rng(7)
    grids = [(11, 4e-6, 2e-5), (1024, f32(8e-6), f32(4e-6)), (512, f32(4e-6), 8e-6), (100, 4e-6, 3e-6),
             (64, 4e-6, 4e-6 * 0.37), (257, 1e-5, 4e-6), (33, 4e-6, 1.3e-5), (1, 4e-6, 1e-6), (2, 4e-6, 4e-6 * 0.5)]
    for nT, dt_o, dt_n in grids:
        sel, nTn = interp_select(nT, dt_o, dt_n, kind)
        assert sel.dtype == np.int32 and sel.shape == (nTn,)
        assert nTn == interp_grid(nT, dt_o, dt_n)[3]                     # same sample count as linear
        assert sel.min() >= 0 and sel.max() <= nT and np.all(np.diff(sel) >= 0)
        t_o = np.arange(0, nT + 1) * dt_o
        t_n = np.arange(1, t_o[-1] // dt_n + 1) * dt_n
        y = np.concatenate([np.zeros((3, 1)), rng.standard_normal((3, nT))], axis=1)   # zero prepended
        want = interpolate.interp1d(t_o, y, axis=1, kind=kind, copy=False, assume_sorted=True)(t_n)
        assert np.array_equal(y[:, sel], want), (kind, nT, dt_o, dt_n)
    assert interp_select(512, f32(4e-6), 8e-6, kind)[1] == 255           # the floor quirk
    with pytest.raises(ValueError):
        interp_select(8, 4e-6, 2e-6, 'cubic')


def test_interpT_rejects_unknown_kinds_before_touching_the_device():
    r"""Every ``kind`` of scipy's ``interp1d`` is served (round 3: the spline kinds too); anything else
    is refused before a device check, and CPU tensors still have no fallback."""
    from mrphy_amd import interp
    rf, gr = torch.zeros(1, 2, 8), torch.zeros(1, 3, 8)
    for kind in ('bogus', 0, 6, 2.5):
        with pytest.raises(NotImplementedError, match='not a kind'):
            interp.interpT(rf, gr, torch.tensor([4e-6]), torch.tensor([2e-6]), kind=kind)
    for kind in ('cubic', 'quadratic', 'slinear', 3):
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            interp.interpT(rf, gr, torch.tensor([4e-6]), torch.tensor([2e-6]), kind=kind)


def test_k0_adjoint_arguments_and_workspace_follow_the_coil_capacity():
    r"""``mrphy_rfgr2beff_bwd``: a multi-coil gradient needs a b1 map, as the forward does (the
    host sums a map-less multi-coil rf into one coil first, ``beffective.py:148-149``) -- rejected on
    the host before any HIP call; the workspace query sizes the one-pass layout (partial sums + packed
    coefficient rows for the padded coil count the launcher picks), and the generic layout beyond 32 coils."""
    lib = mrphy_amd.require_library()
    EINVAL = -1
    N, nM, nT = 1, 64, 32
    for nC in (2, 8, 16, 33):                       # no map, more than one coil
        assert lib.mrphy_rfgr2beff_bwd(0, None, None, None, None, None, None, 0, N, nM, nT, nC, None) == EINVAL
    assert lib.mrphy_rfgr2beff(0, None, 0, None, 0, None, None, 0, 0, None, 0, 0, None, None,
                               N, nM, nT, 4, None) == EINVAL          # the forward's twin check
    groups = 1                                      # bwd_spin_groups(64): ceil(64 / 256)
    ws = lambda nC, dt=0: lib.mrphy_rfgr2beff_bwd_workspace(dt, N, nM, nT, nC)  # noqa: E731
    up = lambda b: (b + 255) // 256 * 256  # noqa: E731
    assert ws(1) == groups * N * 9 * nT * 4
    # 2..32 coils: the partial sums (3 + 2 nC rows of nT per spin group, 256-B rounded) and, behind them, one
    # packed coefficient row of 2 MC + 4 words per spin for the padded coil count MC the launcher picks
    for nC, MC in ((2, 4), (4, 4), (5, 8), (8, 8), (9, 12), (12, 12), (13, 16), (16, 16), (17, 24), (24, 24),
                   (25, 32), (32, 32)):
        assert ws(nC) == up(groups * N * (3 + 2 * nC) * nT * 4) + N * nM * (2 * MC + 4) * 4, nC
    for nC in (33, 64, 70):                                            # blocks of 32 coils: one block's packed rows
        assert ws(nC) == up(groups * N * (3 + 2 * nC) * nT * 4) + N * nM * (2 * 32 + 4) * 4, nC
    assert ws(16, 1) == up(groups * N * 35 * nT * 8) + N * nM * 36 * 8   # fp64


def test_constant_grads_go_to_the_device_where_the_reference_differentiates_them():
    r"""``slowsims.freeprec``, ``beff2ab`` and ``slowsims.blochsim`` are plain autograd in the reference
    (``slowsims.py:86-112,151-174``, ``beffective.py:73-100``): gradients w.r.t. their constants are supplied by
    the kernels (round 3: blochsim; round 4: the other two) -- with CPU tensors every one of them goes on to the
    device check instead of refusing or silently returning none."""
    from mrphy_amd import slowsims, beffective
    M, B = torch.rand(1, 4, 3), torch.rand(1, 4, 8, 3)
    T1 = torch.ones(1, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        slowsims.freeprec(M, torch.tensor(1e-3, requires_grad=True))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        beffective.beff2ab(B, E1=torch.tensor(0.9), E2=torch.tensor(0.8, requires_grad=True))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        slowsims.blochsim(M, B, T1=T1, T2=torch.ones(1, 4))


def test_interp_spline_operator_matches_reference():
    r"""scipy's spline kinds of ``Pulse.interpT`` (``mobjs.py:201,214-215``): the operator
    :func:`mrphy_amd.interp.interp_matrix` takes from scipy, applied in fp64 and rounded once, against
    the reference's own outputs for the config-5 coarse pulse (golden) -- and 'slinear' == 'linear'."""
    import numpy as np
    from mrphy_amd.interp import interp_matrix
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import golden
    I = golden('interp_f32')
    dt_o, dt_n = float(I['coarse_dt'][0]), float(I['dt'][0])
    for kind in ('slinear', 'quadratic', 'cubic'):
        W, n = interp_matrix(1024, dt_o, dt_n, kind)
        assert W.shape == (2048, 1024) and n == 2048
        for ch in ('rf', 'gr'):
            got = (I[f'coarse_{ch}'].astype(np.float64) @ W.T).astype(np.float32)
            assert np.abs(got - I[f'{kind}_{ch}']).max() <= 1e-12, (kind, ch)
    assert np.abs(I['slinear_rf'] - I['rf']).max() <= 1e-12
    with pytest.raises(ValueError):
        interp_matrix(16, 8e-6, 4e-6, 'nearest')
    with pytest.raises(NotImplementedError):
        interp_matrix(1 << 14, 8e-6, 4e-6, 'cubic')


def test_tracked_parity_ledger_is_consistent():
    r"""The tracked ledger of the round's GPU run (profiles/r03_parity.json, written by the GPU suite
    through tests/util.py: record): the run it came from was green, and every distance recorded with a
    bound is within it -- the north star's 1e-5 entries included."""
    import json
    path = os.path.join(ROOT, 'profiles', 'r03_parity.json')
    d = json.load(open(path))
    assert d['meta']['exitstatus'] == 0 and d['meta']['entries'] == len(d['distances']) >= 80
    for k, e in d['distances'].items():
        if 'bound' in e:
            assert e['value'] <= e['bound'], (k, e)
    star = [k for k, e in d['distances'].items() if e.get('bound') == 1e-5]
    assert len(star) >= 30 and any('cfg5_all_spins' in k for k in star) and any('headline_all_spins' in k for k in star)


def test_no_kernel_of_the_library_uses_scratch():
    r"""Round 4: every kernel of the shipped library fits its registers -- no private segment (spills or
    private-memory tables) anywhere, fp64 and parallel-transmit builds included.  Read from the code-object
    metadata of the unit objects (tools/kregs.py)."""
    import importlib.util
    mrphy_amd.build()
    spec = importlib.util.spec_from_file_location('kregs', os.path.join(ROOT, 'tools', 'kregs.py'))
    kregs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kregs)
    ks = kregs.kernels(os.path.join(ROOT, 'mrphy.py_amd', 'build'))
    assert len(ks) > 300
    # (a non-zero vgpr_spill_count with no private segment = values parked in AGPRs of a one-wave-per-SIMD build:
    # register moves, not memory -- four fp64 8-coil builds do that)
    bad = [(n, m['private_segment_fixed_size']) for _, n, m in ks if m['private_segment_fixed_size']]
    assert not bad, bad
    # the fp64 adjoint of blochsim fits two waves per SIMD (<= 256 VGPRs) where the line kernel applies
    k3 = [m['vgpr_count'] for _, n, m in ks if n.startswith('k_bloch_bwd_lines_f64')]
    assert k3 and max(k3) <= 256, k3


@pytest.mark.skipif(not os.path.isdir('/root/reference/mrphy'), reason='reference only in the build container')
def test_gpu_suite_stand_ins_mirror_the_reference_classes():
    r"""The GPU suite replays the object layer on stand-ins (``tests/gpu_common.py``: ``PulseStandIn``,
    ``SpinArrayStandIn``) because the reference cannot travel to the GPU box.  Here, where it imports, they are
    tied to the real classes: constructor / ``to`` signatures (names, kinds, defaults) of ``mobjs.Pulse``, the
    signatures of the methods ``install()`` binds (``Pulse.interpT``, ``SpinArray.applypulse / extract / embed``)
    against the functions bound in their place, and every attribute those functions touch on a real ``Pulse`` /
    ``SpinArray`` against what the stand-ins carry -- so that the stand-ins cannot drift from ``mobjs.py:56-99,
    222-240, 394-450, 512-553`` unnoticed."""
    code = r'''
import sys, inspect, ast, textwrap
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
import torch, mrphy, mrphy_amd
from mrphy import mobjs
import gpu_common as T3

def same_default(a, b):
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        return isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.shape == b.shape and \
            a.dtype == b.dtype and torch.equal(a, b)
    return a == b

def same_signature(f, g, what):
    pf, pg = inspect.signature(f).parameters, inspect.signature(g).parameters
    assert list(pf) == list(pg), (what, list(pf), list(pg))
    for n in pf:
        assert pf[n].kind == pg[n].kind, (what, n, pf[n].kind, pg[n].kind)
        assert (pf[n].default is inspect._empty) == (pg[n].default is inspect._empty), (what, n)
        if pf[n].default is not inspect._empty:
            assert same_default(pf[n].default, pg[n].default), (what, n, pf[n].default, pg[n].default)

# constructor and .to of Pulse (mobjs.py:56-99, 222-240)
same_signature(mobjs.Pulse.__init__, T3.PulseStandIn.__init__, 'Pulse.__init__')
same_signature(mobjs.Pulse.to, T3.PulseStandIn.to, 'Pulse.to')
# the methods install() replaces, against their replacements (mobjs.py:177-220, 394-450, 512-553)
same_signature(mobjs.Pulse.interpT, mrphy_amd._pulse_interpT, 'Pulse.interpT')
same_signature(mobjs.SpinArray.applypulse, mrphy_amd._spinarray_applypulse, 'SpinArray.applypulse')
same_signature(mobjs.SpinArray.extract, mrphy_amd._spinarray_extract, 'SpinArray.extract')
same_signature(mobjs.SpinArray.embed, mrphy_amd._spinarray_embed, 'SpinArray.embed')

def touched(fn, obj='self'):
    """attribute names read or written on `obj` in the source of fn"""
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    return {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == obj}

p, cube = mobjs.Examples.pulse(), mobjs.Examples.spincube()
# what the bound functions touch on the array / the pulse exists on the real objects AND on the stand-ins
arr_attrs = (touched(mrphy_amd._spinarray_applypulse) | touched(mrphy_amd._spinarray_extract) | touched(mrphy_amd._spinarray_embed))
pulse_attrs = touched(mrphy_amd._pulse_interpT) | touched(mrphy_amd._spinarray_applypulse, 'pulse') | touched(mrphy_amd._spinarray_applypulse, 'p')
g = torch.Generator().manual_seed(0)
sp = T3.SpinArrayStandIn(cube.mask, cube.M_, cube.T1_, cube.T2_, cube.γ_)
ps = T3.PulseStandIn(p.rf, p.gr, dt=p.dt)
for a in sorted(arr_attrs):
    assert hasattr(cube, a), ('real SpinCube lacks', a)
    assert hasattr(sp, a), ('SpinArrayStandIn lacks', a)
for a in sorted(pulse_attrs):
    assert hasattr(p, a), ('real Pulse lacks', a)
    assert hasattr(ps, a), ('PulseStandIn lacks', a)
# same values where both exist: the stand-in's constructor normalises as the real one does (mobjs.py:84-99)
for a in ('rf', 'gr', 'dt', 'gmax', 'smax', 'rfmax'):
    x, y = getattr(p, a), getattr(ps, a)
    assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), a
assert (p.desc, p.device, p.dtype) == (ps.desc, ps.device, ps.dtype)
q, qs = p.to(dtype=torch.float64), ps.to(dtype=torch.float64)
assert q.dtype == qs.dtype == torch.float64 and torch.equal(q.rf, qs.rf) and torch.equal(q.dt, qs.dt)
assert ps.to(device=ps.device, dtype=ps.dtype) is ps and p.to(device=p.device, dtype=p.dtype) is p
print('tied', len(arr_attrs), len(pulse_attrs))
''' % (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and 'tied' in out.stdout, (out.stdout[-500:], out.stderr[-2500:])


def test_every_translation_unit_is_built_and_flags_name_real_units():
    r"""The build list (`_lib.UNITS`) names every `csrc/tu_*.hip` and `abi.hip` exactly as often as it has dtype masks,
    no file is forgotten, and the per-unit compiler flags (`_lib.UNIT_FLAGS`) refer to units that exist."""
    import glob
    from mrphy_amd import _lib
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), 'csrc')
    on_disk = {os.path.basename(f) for f in glob.glob(os.path.join(csrc, 'tu_*.hip'))} | {'abi.hip'}
    listed = {src for src, _ in _lib.UNITS}
    assert listed == on_disk, (sorted(listed - on_disk), sorted(on_disk - listed))
    assert len(set(_lib.UNITS)) == len(_lib.UNITS)                      # no (unit, mask) twice
    for key in _lib.UNIT_FLAGS:
        src, mask = key if isinstance(key, tuple) else (key, None)
        assert src in listed, key
        if mask is not None:
            assert (src, mask) in set(_lib.UNITS), key
    # the one-coil float unit of the fused kernel is compiled for the float codes only, with the ILP-first strategy
    assert {m for s_, m in _lib.UNITS if s_ == 'tu_fused_fwd1.hip'} == {_lib._F32, _lib._C64, _lib._P, _lib._PC64}
    assert '-amdgpu-sched-strategy=max-ilp' in _lib.unit_command('tu_fused_fwd1.hip', _lib._F32, '/tmp/x.o')


def test_precision_and_constants_mode_are_context_local():
    r"""``precision(...)`` / ``constants_on(...)`` are contextvars, not module globals: two threads
    inside different ``with`` blocks at the same time each see their own mode (and so pick their own dtype code), the
    main thread sees neither, nested blocks unwind in order, ``precision.set`` moves the process default only."""
    import threading
    assert _host.precision.get() == 'precise' and _host.const_mode_key() == 'rounded'
    seen, inside, go = {}, [threading.Event(), threading.Event()], threading.Event()

    def worker(i, mode, cdev):
        with mrphy_amd.precision(mode), mrphy_amd.constants_on(cdev):
            inside[i].set()
            assert go.wait(10)
            seen[i] = (_host.precision.get(), _host.dtype_code(torch.float32, torch.float32), _host.const_mode_key())
    ts = [threading.Thread(target=worker, args=a) for a in ((0, 'fast', 'cpu'), (1, 'precise', 'native'))]
    [t.start() for t in ts]
    assert all(e.wait(10) for e in inside)                       # both threads are inside their blocks NOW
    assert _host.precision.get() == 'precise' and _host.const_mode_key() == 'rounded'       # ... and we are in neither
    go.set()
    [t.join() for t in ts]
    assert seen == {0: ('fast', _lib.F32, 'cpu'), 1: ('precise', _lib.F32P, 'native')}
    outer = mrphy_amd.precision('fast')
    with outer:
        with mrphy_amd.precision('precise'):
            assert _host.precision.get() == 'precise'
        assert _host.precision.get() == 'fast'
        with outer:                                              # the same object re-entered
            assert _host.precision.get() == 'fast'
        assert _host.precision.get() == 'fast'
    assert _host.precision.get() == 'precise'
    try:
        mrphy_amd.precision.set('fast')
        got = []
        t = threading.Thread(target=lambda: got.append(_host.precision.get()))
        t.start(); t.join()
        assert got == ['fast'] and _host.precision.get() == 'fast'      # the default, also of a new thread
        with mrphy_amd.precision('precise'):
            assert _host.precision.get() == 'precise'
    finally:
        mrphy_amd.precision.set('precise')


def test_grad_workspace_is_device_only_and_context_local():
    from mrphy_amd import workspace
    with pytest.raises(ValueError, match='device memory only'):
        workspace.GradWorkspace((1, 64, 32, 3), torch.float32, 'cpu')
    assert workspace.active() is None


def test_workspace_contexts_are_safe_to_share_between_threads():
    r"""``install(grad_workspace=True)`` keeps ONE process-global ``workspace.auto`` object and every
    routed ``sims.blochsim`` enters it; round 5 kept the ContextVar reset tokens on that shared object, so two threads
    overlapping inside it popped each other's token (``ValueError: Token was created in a different Context``).  The
    tokens now live in a context-local stack: two threads inside the same pool object at the same time, nested and
    re-entered blocks, and the routed function's own local token all unwind cleanly."""
    import threading
    from mrphy_amd import workspace
    pool = workspace.auto()
    errs, inside, go = [], [threading.Event(), threading.Event()], threading.Event()

    def worker(i):
        try:
            with pool:
                inside[i].set()
                assert go.wait(10)
                assert workspace._ACTIVE.get() is pool
                with pool:                                       # re-entered in the same thread
                    assert workspace._ACTIVE.get() is pool
                assert workspace._ACTIVE.get() is pool
            assert workspace._ACTIVE.get() is None and workspace._TOKENS.get() == ()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    assert all(e.wait(10) for e in inside)                       # both threads are inside the SAME object now
    assert workspace.active() is None                            # ... and this thread is not
    go.set()
    [t.join() for t in ts]
    assert errs == []
    # the routed sims.blochsim of install(grad_workspace=True) sets and resets with a local token
    seen = []
    orig, mrphy_amd._AUTO_WS = mrphy_amd._AUTO_WS, pool
    orig_blochsim = mrphy_amd.sims.blochsim
    try:
        mrphy_amd.sims.blochsim = lambda *a, **k: seen.append(workspace._ACTIVE.get())
        rs = [threading.Thread(target=mrphy_amd._blochsim_in_pool) for _ in range(4)]
        [t.start() for t in rs]
        [t.join() for t in rs]
    finally:
        mrphy_amd._AUTO_WS, mrphy_amd.sims.blochsim = orig, orig_blochsim
    assert seen == [pool] * 4 and workspace.active() is None
    # the pool keys its workspaces by thread and serialises building: the attributes that does it with exist
    assert isinstance(pool._lock, type(threading.Lock())) and pool.max_bytes == 64 << 30


def test_sc1_nt_store_carries_its_own_wait_states():
    r"""K0's 16-byte ``global_store_dwordx4 ... nt sc1`` is emitted by inline asm, which LLVM's hazard recogniser
    does not see as a VMEM store -- gfx940+ wants two wait states before a VALU may overwrite the data VGPRs of a store
    wider than 64 bits.  The asm statement therefore ends with ``s_nop 1``: in the disassembly of the K0 units EVERY such
    store is followed by it, wherever the statement was inlined."""
    import tempfile
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump unavailable')
    mrphy_amd.build()
    seen = 0
    for mask in (0x01, 0x02):
        obj = _lib.unit_object(_lib._objdir(), 'tu_rfgr2beff_fwd.hip', mask)
        with tempfile.TemporaryDirectory(prefix='mrphy_co_') as d:
            os.symlink(os.path.abspath(obj), os.path.join(d, 'u.o'))
            if subprocess.run([objdump, '--offloading', 'u.o'], cwd=d, capture_output=True).returncode != 0:
                pytest.skip('llvm-objdump --offloading unavailable')
            co = [f for f in os.listdir(d) if 'gfx950' in f]
            assert len(co) == 1, co
            asm = subprocess.run([objdump, '-d', co[0]], cwd=d, capture_output=True, text=True).stdout.splitlines()
        ins = [ln.split('//')[0].strip() for ln in asm if ln.startswith('\t')]
        for i, x in enumerate(ins):
            if x.startswith('global_store_dwordx4') and 'sc1' in x and 'nt' in x:
                seen += 1
                assert ins[i + 1].startswith('s_nop 1'), (x, ins[i + 1:i + 3])
    assert seen >= 7, seen          # the stores exist (round 4's advisor counted 7 in the float unit alone)
