#!/bin/bash
# configs[4]'s per-GPU shape on a slice of points (D=1000, M=1024, Q=50, free embeddings): per-kernel times
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_c4${1:+_$1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/c4 -o c4 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --N 20000 --D 1000 --M 1024 --Q 50 --regime B --no-cpu-baseline > $O/c4.log 2>&1
tail -1 $O/c4.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
python3 - <<PY
import csv, glob
for f in glob.glob('$O/c4/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    out=['%-70s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']) for r in rows[:12]]
    open('$O/c4_summary.txt','w').write('\n'.join(out)+'\n'); print('\n'.join(out))
PY
rm -f $O/c4/*kernel_trace.csv $O/c4/*agent_info.csv
