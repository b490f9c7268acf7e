#!/bin/bash
# PMC counters of the regime-B (free-embedding) kernels: usage tools/r03_B_pmc.sh <tag> <N> <D> <M> <Q>   (two counter passes + per-kernel times)
set -u
R=${GRAFT_REPO_ROOT:?}
TAG=$1; N=$2; D=$3; M=$4; Q=$5
O=$R/gpurun_out/r03_B_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --N $N --D $D --M $M --Q $Q --regime B --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- python3 $R/bench.py $ARGS > $O/st.log 2>&1
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py $ARGS > $O/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python3 - > $O/summary.txt <<PY
import csv, glob, collections
for f in glob.glob('$O/st/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:10]:
        print('%-70s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
for tag in ('sq1','sq2','sq3'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        agg=collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:44]
            if not any(x in k for x in ('psi2_', 'gen8', 'point_kernel')): continue
            agg.setdefault(k,collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in agg.items():
            print('%-44s %s' % (k, ' '.join('%s=%.4g' % (c, sum(x)/len(x)) for c,x in v.items())))
PY
cat $O/summary.txt
rm -rf $O/*/*kernel_trace.csv $O/*/*agent_info.csv $O/*/*counter_collection.csv
