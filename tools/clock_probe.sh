#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi, 0.25 s period) while a bench command runs: does the FP64 pipe run at the
# 2.4 GHz the 78.6 TFLOP/s peak assumes?   usage (on the GPU box): tools/clock_probe.sh OUT.txt python3 bench.py ...
set -u
OUT=$1; shift
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT.raw 2>&1 &
SAMPLER=$!
"$@" > $OUT.bench 2>&1
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
python3 - <<PY
import re
sclk=[]; pw=[]
for l in open('$OUT.raw'):
    m=re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', l)
    p=re.search(r'Power \(W\): ([\d.]+)', l)
    if m: sclk.append(int(m.group(1)))
    if p: pw.append(float(p.group(1)))
open('$OUT','w').write('samples %d\nsclk MHz: %s\npower W: %s\n' % (len(sclk), sclk, pw))
print(open('$OUT').read()[:3000])
PY
tail -c 600 $OUT.bench
