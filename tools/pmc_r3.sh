#!/bin/bash
# r3 counter passes (separate --pmc runs, kernel trace only): the policy kernel's SQ counters, the task-level physics kernel's SQ counters and HBM bytes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
rm -rf /tmp/pmc_p; timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/pmc_p -- python $R/tools/pmc_policy.py run 4096 > /tmp/pmc_p.log 2>&1
python $R/tools/pmc_policy.py summarize /tmp/pmc_p > $OUT/policy_kernel_pmc.txt 2>&1
rm -rf /tmp/pmc_t1; timeout 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/pmc_t1 -- python $R/tools/pmc_tsc_env.py 8192 > /tmp/pmc_t1.log 2>&1
rm -rf /tmp/pmc_t2; timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_t2 -- python $R/tools/pmc_tsc_env.py 8192 > /tmp/pmc_t2.log 2>&1
rm -rf /tmp/pmc_t3; timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_t3 -- python $R/tools/pmc_tsc_env.py 8192 > /tmp/pmc_t3.log 2>&1
python $R/tools/pmc_tsc_env.py summarize /tmp/pmc_t1 /tmp/pmc_t2 /tmp/pmc_t3 > $OUT/tsc_physics_kernel_pmc.txt 2>&1
tail -2 /tmp/pmc_t1.log >> $OUT/tsc_physics_kernel_pmc.txt
rm -rf /tmp/pmc_t4; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_t4 -- python $R/tools/pmc_tsc_env.py 8192 > /tmp/pmc_t4.log 2>&1
f=$(find /tmp/pmc_t4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 $f | cut -c1-200 >> $OUT/tsc_physics_kernel_pmc.txt
cat $OUT/policy_kernel_pmc.txt; cat $OUT/tsc_physics_kernel_pmc.txt
