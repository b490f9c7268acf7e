# usage: bash tools/prof_cmd.sh <tag> <python script> [args...]   -- rocprofv3 kernel trace + stats into gpurun_out/prof_<tag>
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
TAG=$1; shift
mkdir -p $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 "$@" > $R/gpurun_out/prof_$TAG/run.log 2>&1
cat $R/gpurun_out/prof_$TAG/run.log | tail -15
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/prof_$TAG/*kernel_stats.csv'):
    for i,r in enumerate(csv.DictReader(open(f))):
        if i<25: print('%-70s calls=%s total_us=%.1f avg_us=%.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
