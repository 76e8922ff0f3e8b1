#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/micro/gemm_bench.py > $O/gemm_bench.log 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq1 -o g -- python $R/tools/micro/gemm_bench.py 76800,512,512 3 > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/pmc_sq2 -o g -- python $R/tools/micro/gemm_bench.py 76800,512,512 3 > $O/pmc_sq2.log 2>&1
cat $O/gemm_bench.log; tail -3 $O/pmc_sq1.log $O/pmc_sq2.log
