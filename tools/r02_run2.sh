#!/bin/bash
# quick check: parity tests touching the phase kernels + bench variants
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp_truth.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for v in "GP_P2_ABLATE=1" "GP_P2_VARIANT=4" "GP_P2_VARIANT=0"; do
  echo "== $v" >> $O/ablate.log
  env $v python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['device_ms'])" >> $O/ablate.log
done
cat $O/ablate.log
