#!/bin/bash
# regime B tile-pair kernel: parity tests, then A/B timing at config-3 size
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out/r02_run12
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_partial_terms.py tests/test_gpu_global_step.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r02_run12/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_run12/pytest.log
for v in 0 1; do
  if [ $v == 1 ]; then export GP_B_NOSYM=1; fi
  python3 bench.py --steps 3 --warmup 1 --regime B --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_run12/B_$v.json
  python3 -c "
import json; d=json.load(open('gpurun_out/r02_run12/B_$v.json')); print('nosym=$v', d['ms_per_step'], d['config']['device_ms'])"
done
