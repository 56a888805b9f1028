#!/bin/bash
# r4 GPU call 4: new GPU tests, the task-level teacher with / without the r4 rollout kernels, kernel stats + the launch sequence of one env step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4t
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_policy_chain.py tests/test_tsc_learner.py tests/test_tsc_course_env.py tests/test_episode_means.py tests/test_full_size_properties.py tests/test_gemm_layers.py tests/test_fused_learner.py -m gpu -q > $O/gpu_tests_new.log 2>&1 < /dev/null; echo "pytest rc $?" >> $O/gpu_tests_new.log
tail -12 $O/gpu_tests_new.log
QA_FUSED_POLICY=1 QA_TSC_HYBRID_ACT=1 timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2> $O/tsc1024_new.err < /dev/null | grep '"metric"' > $O/tsc1024_new.json
QA_FUSED_POLICY=1 QA_TSC_HYBRID_ACT=0 timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2> $O/tsc1024_chain_only.err < /dev/null | grep '"metric"' > $O/tsc1024_chain_only.json
timeout 400 python bench.py --tsc --steps 6 --warmup 3 --no_cpu_baseline 2> $O/tsc8192_new.err < /dev/null | grep '"metric"' > $O/tsc8192_new.json
for f in tsc1024_new tsc1024_chain_only tsc8192_new; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["ms_per_step"],2), "ms", {k: round(v*1e3,2) for k,v in d.items() if k.endswith("_s") and isinstance(v,float)})
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
P
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 6 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/tsc1024_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" > $O/tsc1024_env_step_sequence.txt 2>&1
tail -3 $O/tsc1024_env_step_sequence.txt
python $R/tools/top_kernels.py $O/tsc1024_kernel_stats.csv 2>/dev/null | head -30
