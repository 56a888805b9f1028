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
python tools/quick_time.py > $O/quick_time.txt 2>&1
python tools/quick_time.py --terrain >> $O/quick_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmc_f /tmp/pmc_w
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
cd $R
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python tools/pmc_env_step.py 4096 < /dev/null > /tmp/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python tools/pmc_env_step.py 4096 < /dev/null > /tmp/pmc_w.log 2>&1
python tools/pmc_summarize.py /tmp/pmc_f > $O/pmc_fetch.txt 2>&1
python tools/pmc_summarize.py /tmp/pmc_w > $O/pmc_write.txt 2>&1
ls -la $O
