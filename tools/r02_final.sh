#!/bin/bash
# round-2 evidence for the final kernels: bench kernel stats + PMC + traffic, global-step timelines, regime-B kernel stats, potrf phases,
# numbers quoted in DESIGN.md
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
bash tools/r02_prof.sh final > gpurun_out/r02_prof_final.log 2>&1
bash tools/r02_prof_gs.sh > gpurun_out/r02_prof_gs.log 2>&1
O=$R/gpurun_out/r02_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/B -o B --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --regime B --no-cpu-baseline > $O/B.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$O/B/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    out=['%-70s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']) for r in rows[:25]]
    open('$O/B_summary.txt','w').write('\n'.join(out)+'\n'); print('\n'.join(out))
PY
rm -f $O/B/*kernel_trace.csv $O/B/*agent_info.csv
cd $R
tools/ubench/potrf_ubench > $O/potrf_phases.txt 2>&1
tools/ubench/quad_mma_check > $O/quad_mma_check.txt 2>&1
bash tools/r02_numbers.sh > $O/numbers.txt 2>&1
tail -5 $O/numbers.txt
