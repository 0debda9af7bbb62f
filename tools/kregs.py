"""Register / LDS / scratch use of the product kernels, from the assembly hipcc emits:

    python tools/kregs.py [filter]      (compiles csrc/mrphy_hip.hip with --save-temps under /tmp)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
flt = sys.argv[1] if len(sys.argv) > 1 else ''
sys.path.insert(0, ROOT)
d = tempfile.mkdtemp(prefix='kregs_')
import mrphy_amd  # noqa: E402
cmd = [c for c in mrphy_amd._lib.hipcc_command(os.path.join(d, 'x.o')) if c not in ('-shared',)]
cmd.insert(1, '--save-temps')
for f in os.environ.get('KREGS_FLAGS', '').split():
    cmd.insert(1, f)
cmd.insert(1, '-c')
subprocess.run(cmd, check=True, cwd=d, capture_output=True)
s = open(os.path.join(d, 'mrphy_hip-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
names = re.findall(r'\.amdhsa_kernel (\S+)', s)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
for (name, body), dn in zip(re.findall(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S), dem):
    dn = dn.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if flt not in dn:
        continue
    g = lambda k: re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body).group(1)  # noqa: E731
    vg, lds, scr = int(g('next_free_vgpr')), int(g('group_segment_fixed_size')), int(g('private_segment_fixed_size'))
    acc = int(g('accum_offset'))
    waves = min(8, 512 // max(vg, 1)) if vg else 8
    print(f'{dn[:84]:84s} vgpr {vg:4d} (arch {acc:3d}) sgpr {g("next_free_sgpr"):>4} lds {lds:6d} '
          f'scratch {scr:4d}  waves/SIMD<= {waves}')
print('asm:', os.path.join(d, 'mrphy_hip-hip-amdgcn-amd-amdhsa-gfx950.s'))
