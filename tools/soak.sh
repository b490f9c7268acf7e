#!/bin/bash
# Robustness batch on the GPU box (round 4: the double-double product, the asynchronous gp_set_globals and the int8 prototype are new code):
# fuzz against the oracle over shapes and extreme hyper-parameters, bit-identity soaks in both regimes, a leak check, and two resident SCG runs.
#   usage: gpurun --timeout 3000 -- 'tools/soak.sh'      -> gpurun_out/soak.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for s in 1 2 3; do echo "== fuzz parity seed $s"; timeout 600 python tests/devtools/dev_fuzz_parity.py $s 2>&1 | tail -4; done
for s in 1 2; do echo "== fuzz hypers seed $s"; timeout 600 python tests/devtools/dev_fuzz_hypers.py $s 2>&1 | tail -4; done
echo "== fuzz shapes"; timeout 900 python tests/devtools/dev_fuzz_shapes.py 2>&1 | tail -4
echo "== bitwise soak A (bench shape, 1e5 points, 200 reps)"; timeout 600 python tests/devtools/dev_soak_bitwise.py 100000 100 512 10 A 200 2>&1 | tail -3
echo "== bitwise soak A with the int8 phase 1"; GPARML_P1_I8=1 timeout 600 python tests/devtools/dev_soak_bitwise.py 100000 100 512 10 A 100 2>&1 | tail -3
echo "== bitwise soak B (Q = 10)"; timeout 600 python tests/devtools/dev_soak_bitwise.py 20000 20 512 10 B 60 2>&1 | tail -3
echo "== bitwise soak B (Q = 50, M = 1024)"; timeout 900 python tests/devtools/dev_soak_bitwise.py 4000 50 1024 50 B 20 2>&1 | tail -3
echo "== leak check"; timeout 600 python tests/devtools/dev_leak_check.py 2>&1 | tail -4
echo "== resident SCG, Bayesian GPLVM 2e5 x 20, M 128, Q 5, 20 iterations"; timeout 900 python tests/devtools/dev_scg_soak.py 200000 20 128 5 20 2>&1 | tail -4
echo "== resident SCG, benchmark shape with fixed embeddings, 12 iterations"; timeout 900 python tests/devtools/dev_scg_soak.py 1000000 100 512 10 12 fixed 2>&1 | tail -4
} > gpurun_out/soak.txt 2>&1
tail -60 gpurun_out/soak.txt
