#!/bin/bash
# r4 GPU call 12: what moves first when the recorded and the eager update path part ways around iteration 960? (per-tensor policy checksums, lr, std, Adam moments)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4j
timeout 600 python tools/recorded_vs_eager_checksums.py --seeds 3 --iters 975 --keep_rows_around 960 --out gpurun_out/r4j/recorded_vs_eager_diag.json > gpurun_out/r4j/diag.log 2>&1 < /dev/null
tail -1 gpurun_out/r4j/diag.log | cut -c1-700
