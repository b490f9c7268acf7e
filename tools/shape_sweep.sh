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
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 512 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 512 --Q 16
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 512 --Q 30
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 104 --M 512 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 128 --M 512 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 200 --M 512 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 500 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 640 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 100 --M 1024 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 500000 --D 100 --M 1536 --Q 10
[ "${SWEEP_A:-1}" == "1" ] && run --N 1000000 --D 10 --M 256 --Q 5
[ "${SWEEP_A:-1}" == "1" ] && run --N 200000 --D 1000 --M 1024 --Q 50
# free embeddings (regime B): fraction by W_B = N M^2 (4 Q + 10) flops of the whole evaluation
runb() {
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --regime B "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('B %-42s ms %9.3f  frac %.3f  %s' % ('$*', d['ms_per_step'], r['frac'], json.dumps(d['config'].get('kernel_ms', d['config'].get('device_ms')))))"
}
if [ "${SWEEP_B:-1}" == "1" ]; then
runb --N 100000 --D 100 --M 128 --Q 2
runb --N 100000 --D 100 --M 512 --Q 5
runb --N 100000 --D 100 --M 512 --Q 10
runb --N 100000 --D 100 --M 512 --Q 12
runb --N 100000 --D 100 --M 512 --Q 16
runb --N 100000 --D 100 --M 512 --Q 20
runb --N 50000 --D 100 --M 512 --Q 30
runb --N 20000 --D 100 --M 1024 --Q 50
runb --N 20000 --D 100 --M 1024 --Q 60
runb --N 100000 --D 10 --M 128 --Q 10
runb --N 50000 --D 100 --M 300 --Q 10
fi
