#!/bin/bash
# global step rework: linalg + global-step + parity tests, then the kernel timeline
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out/r02_run10
timeout 900 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_global_step.py tests/test_gpu_parity.py tests/test_gpu_hp_truth.py -m gpu -x -q > gpurun_out/r02_run10/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_run10/pytest.log
bash tools/r02_prof_gs.sh 2>&1 | tail -75
