#!/bin/bash
# r4 GPU call 15: config 2 with more envs per GPU (the weak-scaling per-GPU workloads: 8192 / 16384 / 32768 envs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4m
for n in 8192 16384 32768; do
  timeout 300 python bench.py --num_envs $n --no_cpu_baseline --steps 6 --warmup 3 2> gpurun_out/r4m/bench_$n.err < /dev/null | grep '"metric"' > gpurun_out/r4m/bench_cfg2_$n.json
  python - $n <<'P'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/r4m/bench_cfg2_{sys.argv[1]}.json").read()); print(sys.argv[1], round(d["ms_per_step"],1), round(d["value"]), round(d["collection_s"]*1e3,1), round(d["learn_s"]*1e3,1), d["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[1], "failed", e)
P
done
