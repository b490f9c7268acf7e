#!/bin/bash
# kernel stats of the regime-B / embedding-gradient shapes (p2_gen8_kernel and its neighbours).  Output: gpurun_out/r03_gen8[_TAG]/summary.txt
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r03_gen8${1:+_$1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $O/$tag.log 2>&1; }
prof c4 --N 20000 --D 1000 --M 1024 --Q 50 --regime B
prof c2B --N 100000 --D 100 --M 512 --Q 10 --regime B
prof c2B_1e6 --N 1000000 --D 100 --M 512 --Q 10 --regime B
python3 - <<PY
import csv, glob
out=[]
for tag in ('c4','c2B','c2B_1e6'):
    out.append('== ' + tag)
    for f in glob.glob('$O/%s/*kernel_stats.csv' % tag):
        rows=list(csv.DictReader(open(f)))
        out += ['%-64s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']) for r in rows[:14]]
open('$O/summary.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
PY
rm -rf $O/*/*kernel_trace.csv $O/*/*agent_info.csv
cd $R
