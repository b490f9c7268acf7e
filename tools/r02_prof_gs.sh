#!/bin/bash
# kernel-level timeline of the global step (M = 128 and M = 512)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_prof_gs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/c1 -o c1 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --N 100000 --D 10 --M 128 --Q 10 --no-cpu-baseline > $O/c1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/c2.log 2>&1
python3 - <<PY
import csv, glob
for tag in ('c1','c2'):
    for f in glob.glob('$O/%s/*kernel_stats.csv' % tag):
        rows=list(csv.DictReader(open(f)))
        out=['%-70s calls=%5s total_ms=%10.3f avg_us=%10.1f' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3) for r in rows[:45]]
        open('$O/%s_summary.txt' % tag,'w').write('\n'.join(out)+'\n')
    # one evaluation's timeline (last global step): start offsets and durations
    for f in glob.glob('$O/%s/*kernel_trace.csv' % tag):
        rows=list(csv.DictReader(open(f)))
        rows.sort(key=lambda r:int(r['Start_Timestamp']))
        idx=[i for i,r in enumerate(rows) if 'build_kmm' in r['Kernel_Name']]
        i0=idx[-1]
        t0=int(rows[i0]['Start_Timestamp'])
        out=[]
        for r in rows[i0:i0+70]:
            out.append('%9.1f us  dur %8.1f us  %s' % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:80]))
            if 'colsum_kernel' in r['Kernel_Name']: break
        open('$O/%s_timeline.txt' % tag,'w').write('\n'.join(out)+'\n')
        print('\n'.join(out))
PY
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv
