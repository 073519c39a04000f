#!/usr/bin/env python3
"""Per-kernel register / spill / LDS / scratch figures of a gfx950 object or library, from its AMDGPU metadata note
(llvm-readelf --notes).  usage: tools/kernel_resources.py g-nerf_amd/csrc/render.o [name-substring]"""
import os, re, subprocess, sys, tempfile
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
LLVM = '/opt/rocm/lib/llvm/bin/'
tmp = tempfile.mkdtemp()
co = os.path.join(tmp, 'dev.co')
# hipcc objects bundle the device code object: pull out the gfx950 one (a plain code object is read as is)
fat = os.path.join(tmp, 'fat.bin')
subprocess.run([LLVM + 'llvm-objcopy', f'--dump-section=.hip_fatbin={fat}', obj], capture_output=True, text=True)
r = subprocess.run([LLVM + 'clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}', f'--output={co}', '--unbundle'],
                   capture_output=True, text=True)
if r.returncode != 0 or not os.path.exists(co) or not os.path.getsize(co):
    co = obj
txt = subprocess.run([LLVM + 'llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
mangled = re.findall(r'\.name:\s+(\S+)', txt)
plain = subprocess.run(['c++filt'], input='\n'.join(mangled), capture_output=True, text=True).stdout.split('\n')
names = dict(zip(mangled, plain))
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = names.get(g('name'), g('name'))
    if flt not in name:
        continue
    agpr = re.match(r':\s+(\d+)', blk)
    print(f"{name[:100]:100s} vgpr {g('vgpr_count'):>4s} agpr {agpr.group(1) if agpr else '?':>3s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} "
          f"sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}")
