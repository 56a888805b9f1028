#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_tsc_glue.py tests/test_tsc_course_env.py tests/test_tsc_learner.py tests/test_articulated_obstacles.py tests/test_tsc_env.py tests/test_fused_learner.py::test_clip_adam_refuses_to_record_the_pointer_table_path tests/test_gpu_train.py -m gpu -q > $O/gpu_tests.log 2>&1 < /dev/null; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^FAILED|passed|failed" $O/gpu_tests.log | tail -12
timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2> $O/tsc1024.err < /dev/null | grep '"metric"' > $O/tsc1024.json
QA_TSC_GLUE=0 timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2> $O/tsc1024_noglue.err < /dev/null | grep '"metric"' > $O/tsc1024_noglue.json
timeout 400 python bench.py --no_cpu_baseline 2> $O/cfg2.err < /dev/null | grep '"metric"' > $O/cfg2.json
QA_OWN_FWD_NARROW=0 timeout 400 python bench.py --no_cpu_baseline 2> $O/cfg2_nonarrow.err < /dev/null | grep '"metric"' > $O/cfg2_nonarrow.json
for f in tsc1024 tsc1024_noglue cfg2 cfg2_nonarrow; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["ms_per_step"],2), "ms", {k: round(v*1e3,2) for k,v in d.items() if k in ("collection_s","learn_s")})
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
P
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 6 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/tsc1024_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid > $O/tsc1024_env_step_sequence.txt 2>&1
tail -1 $O/tsc1024_env_step_sequence.txt; grep -c "" $O/tsc1024_env_step_sequence.txt
