#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi, 0.25 s period) while a bench command runs: does the FP64 pipe run at the
# 2.4 GHz the 78.6 TFLOP/s peak assumes?  And what does an evaluation cost in joules (mean socket power of the loaded samples x device time)?
#   usage (on the GPU box): tools/clock_probe.sh OUT.txt python3 bench.py ...   (give the bench >= 5 s of timed loop: --steps 300)
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
import json
busy = [(c, p) for c, p in zip(sclk, pw) if pw and p > 0.85 * max(pw)]          # the samples inside the timed loop
line = None
for l in open('$OUT.bench'):
    if l.startswith('{'):
        line = json.loads(l)
summ = ''
if busy and line:
    mp = sum(p for _, p in busy) / len(busy); mc = sum(c for c, _ in busy) / len(busy)
    ms = line['config']['device_ms']['total_ms']
    summ = 'loaded samples %d: mean sclk %.0f MHz, mean socket power %.0f W; %.3f ms device time per evaluation -> %.2f J per evaluation; kernels %s\n' % (
        len(busy), mc, mp, ms, mp * ms * 1e-3, line['config']['device_ms'])
open('$OUT','w').write('samples %d\nsclk MHz: %s\npower W: %s\n%s' % (len(sclk), sclk, pw, summ))
print(open('$OUT').read()[:3000])
PY
tail -c 300 $OUT.bench
