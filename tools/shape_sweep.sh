#!/bin/bash
# Fraction of the FP64 peak over a sweep of fixed-embedding (regime A) shapes around the headline one: finds the shapes that fall off the tuned kernels
# (p1v2_kernel needs D <= 104, p2_fast8_kernel Q <= 11, ...).   usage (through gpurun): tools/shape_sweep.sh > gpurun_out/shape_sweep.txt
cd "$(dirname "$0")/.."
run() {
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']; k = c['device_ms']
print('%-44s ms %8.3f  frac_of_fp64_peak %.3f  psi1 %.3f p1 %.3f p2 %.3f global %.3f  kernel %s' % ('$*', d['ms_per_step'], c['eval_fraction_of_fp64_peak'], k['psi1_ms'], k['p1_kernel_ms'], k['p2_kernel_ms'], k['global_ms'], d['roofline']['kernel']))"
}
run --N 1000000 --D 100 --M 512 --Q 10
run --N 1000000 --D 100 --M 512 --Q 16
run --N 1000000 --D 100 --M 512 --Q 30
run --N 1000000 --D 104 --M 512 --Q 10
run --N 1000000 --D 128 --M 512 --Q 10
run --N 1000000 --D 200 --M 512 --Q 10
run --N 1000000 --D 100 --M 500 --Q 10
run --N 1000000 --D 100 --M 640 --Q 10
run --N 1000000 --D 100 --M 1024 --Q 10
run --N 500000 --D 100 --M 1536 --Q 10
run --N 1000000 --D 10 --M 256 --Q 5
run --N 200000 --D 1000 --M 1024 --Q 50
