#!/bin/bash
# kernel stats of one bench shape: tools/r03_shape_stats.sh TAG <bench.py args>   -> gpurun_out/r03_shape_TAG/summary.txt
set -u
R=${GRAFT_REPO_ROOT:?}
TAG=$1; shift
O=$R/gpurun_out/r03_shape_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/bench.py --no-cpu-baseline --no-extra "$@" > $O/bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra "$@" > $O/st.log 2>&1
python3 - > $O/summary.txt <<PY
import csv, glob, json
for l in open('$O/bench.log'):
    if l.startswith('{'):
        d = json.loads(l); print('bench (not profiled): ms_per_step %.4f  device_ms %s' % (d['ms_per_step'], d['config']['device_ms']))
for f in glob.glob('$O/st/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:30]:
        print('%-64s calls=%5s total_ms=%9.3f avg_us=%9.1f pct=%s' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
cat $O/summary.txt
rm -rf $O/st/*kernel_trace.csv $O/st/*agent_info.csv
