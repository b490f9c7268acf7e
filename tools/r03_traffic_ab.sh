#!/bin/bash
# HBM traffic (FETCH_SIZE) and time of the phase-2 kernel for two library builds on one box: tools/r03_traffic_ab.sh A B  (gparml_amd/lib_A.so.bin ...)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r03_traffic_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  export GPARML_LIB=$R/gparml_amd/lib_$v.so.bin
  rm -rf $O/$v
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$v -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/$v.log 2>&1
  python3 - <<PY
import csv, glob, json
vals=[float(r['Counter_Value']) for f in glob.glob('$O/$v/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'p2_fast8' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
b=[json.loads(l) for l in open('$O/$v.log') if l.startswith('{')][-1]
print('$v rep $rep: FETCH_SIZE raw KB %.4g (x2 x1024 = %.2f GB)  p2_kernel_ms %.3f' % (sum(vals)/len(vals), 2*1024*sum(vals)/len(vals)/1e9, b['config']['device_ms']['p2_kernel_ms']))
PY
done; done
