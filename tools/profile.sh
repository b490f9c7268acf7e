#!/bin/bash
# rocprofv3 profile of a bench.py workload on the GPU box: kernel stats, PMC passes (each its own run: counters are never combined with a trace
# domain other than --kernel-trace), HBM traffic of the headline kernels.  Replaces the per-round one-offs (r02_prof.sh, r03_prof.sh, r03_B_pmc.sh ...;
# they are in the git history).
#   usage (through gpurun):  tools/profile.sh TAG [--filter 'name1|name2'] [-- bench.py arguments]
#   examples:  tools/profile.sh r04_bench                                                  the default headline workload (+ traffic.json)
#              tools/profile.sh r04_B_c4 --filter 'psi2_|gen8|point' -- --N 20000 --D 1000 --M 1024 --Q 50 --regime B
# Output: gpurun_out/prof_TAG/summary.txt (+ traffic.json for the headline workload); copy what is to be kept into profiles/.
set -u
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
TAG=${1:?tag}; shift
FILTER='p1_kernel|p1v2_kernel|p1i8|p2_kernel|p2_fast|p2_gen8|psi1_kernel|psi2_|ddacc|solve_residual|potrf'
if [ "${1:-}" == "--filter" ]; then FILTER=$2; shift 2; fi
if [ "${1:-}" == "--" ]; then shift; fi
ARGS="--no-cpu-baseline --no-extra $*"
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $ARGS > $O/stats.log 2>&1
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS > $O/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
run write WRITE_SIZE
FILTER="$FILTER" python3 - <<PY
import csv, collections, glob, json, datetime, os, re
flt = re.compile(os.environ['FILTER'])
out = []
for f in glob.glob('$O/stats/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    out += ['%-70s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']) for r in rows[:40]]
out.append('')
for tag in ('sq1', 'sq2', 'sq3', 'fetch', 'write'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:44]
            if not flt.search(k):
                continue
            agg.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for k, v in agg.items():
            out.append('%-44s %s' % (k, ' '.join('%s=%.4g' % (c, sum(x) / len(x)) for c, x in v.items())))
lines = [json.loads(l) for l in open('$O/stats.log') if l.startswith('{')]
bench = lines[-1] if lines else None
if bench:
    out.append('')
    out.append('bench line of the profiled command: ' + json.dumps({k: bench[k] for k in ('value', 'ms_per_step', 'roofline')}))
open('$O/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
def avg(tag, counter, key):
    vals = [float(r['Counter_Value']) for f in glob.glob('$O/%s/*counter_collection.csv' % tag) for r in csv.DictReader(open(f)) if key in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return sum(vals) / len(vals) if vals else None
if bench and bench['config'].get('regime') == 'A':
    tr = {}
    kname = bench['roofline']['kernel']                       # e.g. gp::p2_fast8_kernel<3>
    for key, name in ((kname.split('::')[-1].split('<')[0], 'p2_kernel'), ('p1v2_kernel', 'p1_kernel'), ('psi1_kernel', 'psi1_kernel')):
        f = avg('fetch', 'FETCH_SIZE', key); w = avg('write', 'WRITE_SIZE', key)
        if f is not None and w is not None:
            # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads -> x2 (MI355X_MICROARCH.md, HBM)
            tr[name + '_hbm_bytes_per_launch'] = (2.0 * f + w) * 1024.0; tr[name + '_fetch_kb_raw'] = f; tr[name + '_write_kb_raw'] = w
    c = bench['config']
    tr['kernel'] = kname; tr['N'] = c['N_per_gpu']; tr['D'] = c['D']; tr['M'] = c['M']; tr['Q'] = c['Q']
    tr['date'] = datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ')
    tr['command'] = 'tools/profile.sh $TAG (bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'
    json.dump(tr, open('$O/traffic.json', 'w'), indent=1)
    print(tr)
PY
rm -rf $O/*/*kernel_trace.csv $O/*/*agent_info.csv $O/*/*counter_collection.csv
cd $R
