cd $GRAFT_REPO_ROOT
{
for m in "A 1000 7 130 10" "Ae 1000 7 130 10" "B 1000 7 130 10" "B 600 3 512 10" "B 400 2 70 20"; do echo "=== probe $m"; timeout 300 python3 tests/devtools/dev_poison_probe.py $m 2>&1 | tail -45; done
} > gpurun_out/r06_poison_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_wide_latent.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06_wide_latent.txt
tools/ab.sh -r 3 -a "--regime B --N 100000 --steps 3 --warmup 1" intree vtabp vtabs > gpurun_out/r06_ab_vtab.txt 2>&1
tail -50 gpurun_out/r06_poison_probe.txt; cat gpurun_out/r06_wide_latent.txt; cat gpurun_out/r06_ab_vtab.txt
