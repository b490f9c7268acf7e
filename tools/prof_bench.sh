# rocprofv3 kernel-trace + stats of the default bench command; summary copied to gpurun_out/prof_bench
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
mkdir -p $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench/run.log 2>&1
tail -2 $R/gpurun_out/prof_bench/run.log
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/prof_bench/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    out=['%-64s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']) for r in rows[:30]]
    open('$R/gpurun_out/prof_bench/summary.txt','w').write('\n'.join(out)+'\n')
    print('\n'.join(out[:14]))
PY
rm -f $R/gpurun_out/prof_bench/*kernel_trace.csv
