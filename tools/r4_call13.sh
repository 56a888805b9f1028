#!/bin/bash
# r4 GPU call 13: does EACH update path repeat itself run to run over 975 iterations (every checksum field incl. per-tensor policy sums and Adam moments)?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4k
timeout 900 python tools/recorded_vs_eager_checksums.py --seeds 3 --iters 20 --eager_twice_iters 975 --out gpurun_out/r4k/run_to_run_975.json > gpurun_out/r4k/run.log 2>&1 < /dev/null
tail -1 gpurun_out/r4k/run.log | cut -c1-900
