#!/bin/bash
# r4 GPU call 8: the full GPU test suite at HEAD, then the round-end measurement pass
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/final/gpu_tests.log 2>&1 < /dev/null
tail -3 gpurun_out/final/gpu_tests.log
bash tools/final_measure.sh > gpurun_out/final/final_measure.log 2>&1
cat gpurun_out/final/bench_cfg2.json | head -c 600
