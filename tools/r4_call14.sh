#!/bin/bash
# r4 GPU call 14: SQ counter pass of the LDS-DMA GEMM kernel beside the register-staged one (forward / dX / dW of the 24,576 x 672 x 512 layer)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4l
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_gd /tmp/pmc_gd2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_gd -- python $R/tools/pmc_gemm_dma.py > /tmp/pmc_gd.log 2>&1 < /dev/null
python $R/tools/pmc_gemm_dma.py summarize /tmp/pmc_gd > $R/gpurun_out/r4l/gemm_dma_pmc.txt 2>&1
tail -3 /tmp/pmc_gd.log >> $R/gpurun_out/r4l/gemm_dma_pmc.txt
head -50 $R/gpurun_out/r4l/gemm_dma_pmc.txt
