"""Register / LDS / scratch use of every kernel of the library, read from the code-object metadata of the
unit objects `mrphy_amd.build()` leaves under mrphy.py_amd/build/ (no recompilation):

    python tools/kregs.py [filter] [--scratch] [--objdir DIR]

`--scratch` lists only kernels with a private segment (spills); the exit code is then the number found.
`--objdir tools/build_dev` reads the development build instead.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LLVM = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'lib', 'llvm', 'bin')


def kernels(objdir):
    r"""[(object, demangled kernel name, {metadata})] for every kernel of every unit object in `objdir`."""
    out = []
    with tempfile.TemporaryDirectory(prefix='kregs_') as d:
        for obj in sorted(glob.glob(os.path.join(objdir, '*.o'))):
            b = os.path.basename(obj)
            os.symlink(os.path.abspath(obj), os.path.join(d, b))
            subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', b], cwd=d, check=True,
                           capture_output=True)
            for co in glob.glob(os.path.join(d, b + '.*gfx950')):
                txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], check=True,
                                     capture_output=True, text=True).stdout
                for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
                    g = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk).group(1)  # noqa: E731
                    out.append((b, g('name'), {k: int(g(k)) for k in (
                        'vgpr_count', 'sgpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size',
                        'vgpr_spill_count')}))
    names = subprocess.run(['c++filt'] + [k[1] for k in out], capture_output=True, text=True).stdout.splitlines()
    return [(o, n.replace('(anonymous namespace)::', '').replace('mrphy::', '').replace('void ', '').split('(')[0], m)
            for (o, _, m), n in zip(out, names)]


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    objdir = os.path.join(ROOT, 'mrphy.py_amd', 'build')
    if '--objdir' in sys.argv:
        objdir = sys.argv[sys.argv.index('--objdir') + 1]
        args = [a for a in args if a != objdir]
    flt = args[0] if args else ''
    only_scratch = '--scratch' in sys.argv
    n = 0
    ks = kernels(objdir)
    for obj, name, m in ks:
        if flt not in name or (only_scratch and not m['private_segment_fixed_size']):
            continue
        n += 1
        vg = m['vgpr_count']
        waves = min(8, 512 // (-(-vg // 8) * 8)) if vg else 8        # VGPRs are allocated in blocks of 8
        if '--brief' in sys.argv:
            print(f'{name[:70]:70s} v{vg:4d} lds{m["group_segment_fixed_size"]:6d} scr{m["private_segment_fixed_size"]:5d} spill{m["vgpr_spill_count"]:4d}')
            continue
        print(f'{name[:100]:100s} vgpr {vg:4d} sgpr {m["sgpr_count"]:4d} lds {m["group_segment_fixed_size"]:6d} '
              f'scratch {m["private_segment_fixed_size"]:5d} spilled {m["vgpr_spill_count"]:4d}  waves/SIMD<= {waves}  [{obj}]')
    print(f'{n} of {len(ks)} kernels listed')
    sys.exit(n if only_scratch else 0)
