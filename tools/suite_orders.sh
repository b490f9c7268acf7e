#!/bin/bash
# Round 6's hunt for the one unreproduced parity failure (tests/test_gpu_tile_phase2.py, round 5): the failing CONDITION.
#   1. tools/stress_inproc.py: the 18 forced-tile shapes REPS times in ONE long-lived process, contexts of other shapes and the jitter path interleaved,
#      every result compared with the oracle and bit for bit with the first run of its shape;
#   2. the whole GPU suite with its files in reversed and in shuffled order (GPARML_TEST_ORDER, tests/conftest.py).
#   usage (GPU box): tools/suite_orders.sh [REPS]    -> gpurun_out/suite_orders.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPS=${1:-200}
{
echo "== in-process stress, $REPS repetitions of 18 shapes, tile kernel forced"
GPARML_B_PHASE2=tiles timeout 2400 python3 tools/stress_inproc.py $REPS 2>&1 | tail -15
echo "== GPU suite, files in reversed order"
GPARML_TEST_ORDER=reversed timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8
echo "== GPU suite, files shuffled (seed 6)"
GPARML_TEST_ORDER=shuffle:6 timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8
} > gpurun_out/suite_orders.txt 2>&1
tail -40 gpurun_out/suite_orders.txt
