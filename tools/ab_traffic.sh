#!/bin/bash
# Same-box A/B of library builds on the headline workload WITH the HBM traffic of the headline kernels (FETCH_SIZE / WRITE_SIZE passes of rocprofv3,
# one counter per run, never combined with a trace domain other than --kernel-trace).
#   usage (through gpurun): tools/ab_traffic.sh [-r ROUNDS] NAME ...      NAME as in tools/ab.sh ("intree" or lib_NAME.so.bin)
set -u
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd $R
ROUNDS=3
while getopts "r:" o; do case $o in r) ROUNDS=$OPTARG;; esac; done
shift $((OPTIND - 1))
tools/ab.sh -r $ROUNDS "$@"
for spec in "$@"; do
  v=${spec%%:*}
  lib=$R/gparml_amd/lib_$v.so.bin; [ "$v" == "intree" ] && lib=$R/gparml_amd/libgparml_hip.so
  export GPARML_LIB=$lib
  # run-time switches of the variant (NAME:ENV=VALUE[,ENV=VALUE]) are exported into this shell: rocprofv3 must start python3 itself (no env / bash -c hop)
  if [ "$spec" != "$v" ]; then for kv in $(echo "${spec#*:}" | tr ',' ' '); do export "$kv"; done; fi
  tag=$(echo "$spec" | tr ':=,' '___')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/abt_${tag}_$ctr; rm -rf $O; mkdir -p $O
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $ctr -d $O -o c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/log.txt 2>&1)
  done
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('gpurun_out/abt_%s_%s/*counter_collection.csv' % (v, ctr)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if any(s in k for s in ('p2_fast8', 'p1v2_kernel<', 'psi1_kernel')):
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    f = sum(d['FETCH_SIZE']) / max(1, len(d['FETCH_SIZE'])); w = sum(d['WRITE_SIZE']) / max(1, len(d['WRITE_SIZE']))
    # KB units; gfx950 FETCH_SIZE counts half the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md): x2
    print('%-8s %-40s fetch_kb %.0f write_kb %.0f  hbm_GB_per_launch %.2f' % (v, k[:40], f, w, (2 * f + w) * 1024 / 1e9))
PY
  rm -rf gpurun_out/abt_${tag}_*/*kernel_trace.csv gpurun_out/abt_${tag}_*/*counter_collection.csv gpurun_out/abt_${tag}_*/*agent_info.csv
  if [ "$spec" != "$v" ]; then for kv in $(echo "${spec#*:}" | tr ',' ' '); do unset "${kv%%=*}"; done; fi
done
