#!/bin/bash
# r4 GPU call 10: recorded vs eager checksums (state vs read-outs, run-to-run determinism), then the full GPU suite and the measurement pass at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4h gpurun_out/final
timeout 900 python tools/recorded_vs_eager_checksums.py --seeds 3 --iters 1000 --eager_twice_iters 60 --out gpurun_out/r4h/recorded_vs_eager.json > gpurun_out/r4h/recorded_vs_eager.log 2>&1 < /dev/null
tail -2 gpurun_out/r4h/recorded_vs_eager.log
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/final/gpu_tests.log 2>&1 < /dev/null
grep -a "passed\|failed" gpurun_out/final/gpu_tests.log | tail -2
bash tools/final_measure.sh > gpurun_out/final/final_measure.log 2>&1
head -c 400 gpurun_out/final/bench_cfg2.json
