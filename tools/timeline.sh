#!/bin/bash
# Kernel-level timeline of ONE whole evaluation (every launch: start offset, duration, gap to the previous launch) from rocprofv3 --kernel-trace.
#   usage (through gpurun):  tools/timeline.sh TAG [bench.py arguments]      e.g.  tools/timeline.sh r05_config1 --N 100000 --D 10 --M 128 --Q 10
# The evaluation shown is the LAST one of the run (it starts at the last launch of the Kmm / Zaug build of gp_set_globals ...
# ... and ends with the last kernel before the next host read-back).  Output: gpurun_out/timeline_TAG/timeline.txt
set -u
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
TAG=${1:?tag}; shift
O=$R/gpurun_out/timeline_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra "$@" > $O/bench.log 2>&1
python3 - <<PY
import csv, glob, json
for f in glob.glob('$O/t/*kernel_trace.csv'):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # one evaluation = from the Zaug build of gp_set_globals (zaug_kernel) to the gradient read-back of gp_finish (finish_kernel + its copy):
    # take the last complete one
    st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
    starts = [i for i, r in enumerate(rows) if 'zaug_kernel' in r['Kernel_Name']]
    ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('finish_kernel') or r['Kernel_Name'].startswith('finish_kernel')]
    b = ends[-1] + 1
    a = max(i for i in starts if i < b)
    t0 = st[a]
    out = ['# %s: %d launches, first start -> last end %.1f us, sum of durations %.1f us' % ('$TAG', b - a, (en[b - 1] - t0) / 1e3, sum(en[i] - st[i] for i in range(a, b)) / 1e3)]
    for i in range(a, b):
        out.append('%9.1f us  dur %8.1f us  gap %6.1f us  %s' % ((st[i] - t0) / 1e3, (en[i] - st[i]) / 1e3, (st[i] - en[i - 1]) / 1e3 if i > a else 0.0, rows[i]['Kernel_Name'][:110]))
    lines = [l for l in open('$O/bench.log') if l.startswith('{')]
    if lines:
        bj = json.loads(lines[-1])
        out.append('# bench line of the traced command: ms_per_step %.4f, device_ms %s' % (bj['ms_per_step'], json.dumps(bj['config'].get('device_ms'))))
    open('$O/timeline.txt', 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))
PY
rm -f $O/t/*kernel_trace.csv $O/t/*agent_info.csv
