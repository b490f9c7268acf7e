#!/bin/bash
# VGPRs / scratch / occupancy of every kernel in one source: tools/kernel_resources.sh gparml_amd/csrc/X.hip [filter]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p /tmp/asm && cd /tmp/asm
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$ROOT/$1" -save-temps=obj -o /tmp/asm/res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
name = None; vals = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = m.group(1); vals[name] = {}
    for k in ('VGPRs', 'AGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]', 'SGPRs'):
        m = re.search(r' ' + k + r': (\d+)', line)
        if m and name: vals[name][k.split(' ')[0]] = m.group(1)
flt = '$2'
for n, v in vals.items():
    if flt in n: print('%-70s %s' % (n[:70], ' '.join('%s=%s' % kv for kv in v.items())))
"
