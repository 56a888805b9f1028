#!/bin/bash
# r4 GPU call: GPU test suite, LDS-DMA GEMM microbenchmark, end-to-end A/B of the learner's GEMM routing (every command under its own timeout)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 < /dev/null; echo "pytest rc $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 600 python tools/gemm_dma_bench.py --json $O/gemm_dma_bench.json > $O/gemm_dma_bench.log 2>&1 < /dev/null
tail -3 $O/gemm_dma_bench.log | cut -c1-400
timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_default.err < /dev/null | grep '"metric"' > $O/bench_default.json
QA_OWN_LAYERS=all QA_GEMM_DMA=0 timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_allown_reg.err < /dev/null | grep '"metric"' > $O/bench_allown_reg.json
QA_OWN_LAYERS=all QA_GEMM_DMA=1 timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_allown_dma.err < /dev/null | grep '"metric"' > $O/bench_allown_dma.json
QA_OWN_LAYERS=all QA_GEMM_DMA=1 QA_PAD_K=0 timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_allown_dma_nopad.err < /dev/null | grep '"metric"' > $O/bench_allown_dma_nopad.json
QA_LEAN_EXPORTS=0 timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_default_nolean.err < /dev/null | grep '"metric"' > $O/bench_default_nolean.json
for f in default allown_reg allown_dma allown_dma_nopad default_nolean; do python - <<P
import json
try:
    d=json.load(open("$O/bench_$f.json")); print("$f", round(d["ms_per_step"],2), "ms  rollout", round(d["collection_s"]*1e3,2), "learn", round(d["learn_s"]*1e3,2), "env kernel us", round(d["roofline"]["kernel_ms"]*1e3,1))
except Exception as e: print("$f", "FAILED", e)
P
done
