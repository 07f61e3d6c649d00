"""Per-kernel register / spill / LDS table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py psnerf_amd/csrc/mlp_infer.hip [name filter]"""
import re
import subprocess
import sys
import tempfile

import os
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-c', src,
                        '-o', d + '/o.o', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, cwd=d)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(['c++filt', c['name']], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print('%-90s vgpr %4s agpr %4s spill %3s sgpr %4s lds %6s occ %s' % (name[:90], c.get('VGPRs'), c.get('AGPRs'), c.get('VGPRs Spill'),
                                                                      c.get('SGPRs'), c.get('LDS Size [bytes/block]'), c.get('Occupancy [waves/SIMD]')))
