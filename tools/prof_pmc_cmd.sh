# usage: bash tools/prof_pmc_cmd.sh <tag> <kernel-substring> <python script> [args...]  -- SQ counters per kernel (separate passes)
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
TAG=$1; KEY=$2; shift; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc $PMC -d $O/$tag -o $tag --output-format csv -- python3 "$@" > $O/$tag.log 2>&1; }
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" run sq1 "$@"
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" run sq2 "$@"
PMC="FETCH_SIZE" run fetch "$@"
python3 - <<PY
import csv, collections, glob
for tag in ('sq1','sq2','fetch'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        agg=collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if '$KEY' not in r['Kernel_Name']: continue
            agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        print(tag, ' '.join('%s=%.4g' % (c, sum(x)/len(x)) for c,x in agg.items()))
PY
rm -rf $O/*/*kernel_trace.csv $O/*/*counter_collection.csv
