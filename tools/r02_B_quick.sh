#!/bin/bash
# regime B: parity tests + the config-3-size bench with per-kernel stats
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_global_step.py tests/test_gpu_partial_terms.py -m gpu -x -q 2>&1 | tail -3
O=$R/gpurun_out/r02_B_quick
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/B -o B --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --regime B --no-cpu-baseline > $O/B.log 2>&1
tail -1 $O/B.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
python3 - <<PY
import csv, glob
for f in glob.glob('$O/B/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    print('\n'.join('%-60s calls=%4s avg_us=%10.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3) for r in rows[:4]))
PY
rm -f $O/B/*kernel_trace.csv $O/B/*agent_info.csv
