#!/bin/bash
# The GPU suite with every allocation and every per-evaluation buffer poisoned (GPARML_POISON=1, csrc/gp_common.h), all failures listed (no -x).
#   usage (GPU box): tools/poison_suite.sh [pytest arguments]   -> gpurun_out/poison_suite.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== tests/test_gpu_poison.py (poison switched on inside a child process)"
timeout 1200 python -m pytest tests/test_gpu_poison.py -m gpu -q 2>&1 | tail -15
echo "== whole GPU suite under GPARML_POISON=1"
GPARML_POISON=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" 2>&1 | tail -60
} > gpurun_out/poison_suite.txt 2>&1
tail -80 gpurun_out/poison_suite.txt
