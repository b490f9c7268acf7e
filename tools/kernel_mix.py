#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's -save-temps assembly, and the issue-time bound it implies on the shared FP64 VALU / MFMA pipe
(one v_mfma_f64_4x4x4_4b = 16 pipe cycles, one wave64 VALU instruction = 4; measured issue intervals 17 / 4.3).
For a kernel whose body runs once per work item and wave (psi2_tile_kernel: one point and tile per pass through both branches) this is the
per-item instruction count: compare pipe_cycles with the measured cycles per item.
usage: tools/kernel_mix.py gparml_amd/csrc/psi2_tile.hip 'psi2_tile_kernelILi52E'"""
import collections
import os
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp(prefix='kmix_')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', os.path.abspath(src), '-save-temps=obj', '-o', tmp + '/k.o'],
                      cwd=tmp, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')][0]
txt = open(os.path.join(tmp, asm)).read()
m = re.search(r'^(\w*%s\w*):\s*(?:;[^\n]*)?\n(.*?)s_endpgm' % re.escape(pat), txt, re.S | re.M)
if not m:
    raise SystemExit('kernel matching %r not found' % pat)
cls, ops = collections.Counter(), collections.Counter()
for ln in m.group(2).split('\n'):
    ln = ln.strip()
    if not ln or ln[0] in ';.' or ln.endswith(':'):
        continue
    op = ln.split()[0]
    key = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'waitcnt' if op.startswith('s_waitcnt')
           else 'salu' if op.startswith('s_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
    cls[key] += 1
    ops[op] += 1
f64 = sum(v for k, v in ops.items() if k.startswith('v_') and 'f64' in k and not k.startswith('v_mfma'))
print(m.group(1))
print(dict(cls), ' FP64 VALU %d, other VALU %d' % (f64, cls['valu'] - f64))
print('pipe cycles (MFMA x 16 + VALU x 4): %d   at the measured issue intervals (17 / 4.3): %d' % (16 * cls['mfma'] + 4 * cls['valu'], 17 * cls['mfma'] + 4.3 * cls['valu']))
print('most frequent:', ops.most_common(16))
