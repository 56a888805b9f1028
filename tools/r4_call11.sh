#!/bin/bash
# r4 GPU call 11: config 2 return curves on r4's code (12 seeds, the r3 protocol), then recorded-vs-eager state checksums for two more seeds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4i
timeout 900 python tools/return_curve_parity.py --side gpu --num_envs 1024 --iters 1000 --seeds 1 2 3 4 5 6 7 8 9 10 11 12 --out gpurun_out/r4i/hip_cfg2_12seeds_r4.json > gpurun_out/r4i/hip_cfg2.log 2>&1 < /dev/null
tail -2 gpurun_out/r4i/hip_cfg2.log
timeout 900 python tools/recorded_vs_eager_checksums.py --seeds 1 6 --iters 1000 --out gpurun_out/r4i/recorded_vs_eager_s1_s6.json > gpurun_out/r4i/recorded_vs_eager.log 2>&1 < /dev/null
tail -2 gpurun_out/r4i/recorded_vs_eager.log | cut -c1-600
