cd /tmp && export TMPDIR=/tmp
set -eu
R=${GRAFT_REPO_ROOT:?}
mkdir -p $R/gpurun_out/prof_gemm
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_gemm/p1 -o p1 --output-format csv -- python3 $R/tools/dev_gemm_bench.py > $R/gpurun_out/prof_gemm/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/prof_gemm/p2 -o p2 --output-format csv -- python3 $R/tools/dev_gemm_bench.py > $R/gpurun_out/prof_gemm/p2.log 2>&1
ls -R $R/gpurun_out/prof_gemm | head -30
