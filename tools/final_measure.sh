#!/bin/bash
# Round-end measurement pass on the GPU box: bench lines (configs 2, 3, terrain), rocprofv3 kernel stats of the default
# bench command, PMC FETCH_SIZE / WRITE_SIZE passes of the env-step kernel (separate passes, kernel-trace only).
# Outputs land in gpurun_out/final/ (small CSV/JSON only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 400 python bench.py --amp --no_cpu_baseline > $O/bench_cfg3_amp.json 2> $O/bench_cfg3.err
timeout 400 python bench.py --terrain trimesh --no_cpu_baseline > $O/bench_cfg2_trimesh.json 2> $O/bench_trimesh.err
( export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 QA_FORCE_DATA_PARALLEL=1; timeout 400 python bench.py --no_cpu_baseline > $O/bench_cfg2_dp_path_1gpu.json 2> $O/bench_dp.err )
python tools/policy_time.py 4096 > $O/policy_time.txt 2>&1
python tools/policy_time.py 16384 >> $O/policy_time.txt 2>&1
python tools/mlp_profile.py 4096 > $O/mlp_profile.txt 2>&1
python tools/quick_time.py > $O/quick_time.txt 2>&1
python tools/quick_time.py --terrain >> $O/quick_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmc_f /tmp/pmc_w
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
rm -rf /tmp/prof
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline --amp < /dev/null > /tmp/prof_amp.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_cfg3_amp_kernel_stats.csv
rm -rf /tmp/prof
timeout 700 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline --steps 4 < /dev/null > /tmp/prof_tr.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/gap_report.py "$f" 0.7 > $O/gap_report.txt 2>&1 && python $R/tools/step_sequence.py "$f" > $O/step_sequence.txt 2>&1
cd $R
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python tools/pmc_env_step.py 4096 < /dev/null > /tmp/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python tools/pmc_env_step.py 4096 < /dev/null > /tmp/pmc_w.log 2>&1
python tools/pmc_summarize.py /tmp/pmc_f > $O/pmc_fetch.txt 2>&1
python tools/pmc_summarize.py /tmp/pmc_w > $O/pmc_write.txt 2>&1
ls -la $O
