cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "B" 2>&1 | tail -8 > gpurun_out/r06_parity_b.txt
cat gpurun_out/r06_parity_b.txt
for q in 10 12 14 16; do
  tools/ab.sh -r 2 -a "--regime B --N 100000 --Q $q --steps 3 --warmup 1" intree intree:GPARML_B_SYM_MAXQ=10 2>&1 | grep -v amdgpu.ids | sed "s/^/Q=$q /"
done > gpurun_out/r06_ab_sym_q.txt
cat gpurun_out/r06_ab_sym_q.txt
tools/poison_suite.sh > /dev/null 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/poison_suite.txt | cut -c1-150 | head -80
