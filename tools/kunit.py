"""Compile ONE unit of the library with extra flags into a scratch directory and print its kernels' registers:

    python tools/kunit.py tu_blochsim_fwd.hip 0x08 [filter] [-DNAME=VALUE ...] [--asm]

(0x08 = dtype mask, `-` for the units without one.)  `--asm` also leaves the device assembly in the scratch
directory (hipcc --save-temps) and prints its path.
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mrphy_amd  # noqa: E402,F401
from mrphy_amd import _lib  # noqa: E402
import kregs  # noqa: E402

src, mask = sys.argv[1], sys.argv[2]
rest = sys.argv[3:]
extra = [a for a in rest if a.startswith('-') and a != '--asm']
flt = next((a for a in rest if not a.startswith('-')), '')
d = os.environ.get('KUNIT_DIR', '/tmp/kexp')
os.makedirs(d, exist_ok=True)
for f in os.listdir(d):
    if f.endswith(('.o', '.d', '.s', '.bc', '.hipfb', '.hipi', '.out')) or 'gfx950' in f:
        os.remove(os.path.join(d, f))
m = None if mask == '-' else int(mask, 0)
obj = _lib.unit_object(d, src, m)
cmd = _lib.unit_command(src, m, obj, extra + (['--save-temps'] if '--asm' in rest else []))
r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:])
    sys.exit(1)
for o, name, md in kregs.kernels(d):
    if flt in name:
        print(f'{name[:96]:96s} vgpr {md["vgpr_count"]:4d} sgpr {md["sgpr_count"]:4d} lds {md["group_segment_fixed_size"]:6d} '
              f'scratch {md["private_segment_fixed_size"]:5d} spilled {md["vgpr_spill_count"]:4d}')
if '--asm' in rest:
    print([os.path.join(d, f) for f in os.listdir(d) if f.endswith('.s') and 'gfx950' in f])
