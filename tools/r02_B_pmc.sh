#!/bin/bash
# PMC counters of the Q >= 25 pair kernels on the config-3 size, free embeddings (two passes)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_B_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --regime B --no-cpu-baseline > $O/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
python3 - <<PY
import csv, glob, collections
for tag in ('sq1','sq2'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        agg=collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:44]
            if 'psi2_' not in k: continue
            agg.setdefault(k,collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in agg.items():
            print('%-44s %s' % (k, ' '.join('%s=%.4g' % (c, sum(x)/len(x)) for c,x in v.items())))
PY
rm -rf $O/*/*kernel_trace.csv $O/*/*agent_info.csv
