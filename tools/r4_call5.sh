#!/bin/bash
# r4 GPU call 5: the tests that failed in calls 3 / 4, recorded-vs-eager checksums (fixed protocol), the teacher after the hybrid-act kernel's out-of-place roll
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_tsc_course_env.py tests/test_tsc_learner.py tests/test_full_size_properties.py tests/test_gemm_layers.py tests/test_fused_learner.py tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_self_collision.py tests/test_articulated_obstacles.py -m gpu -q > $O/gpu_tests.log 2>&1 < /dev/null; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^FAILED|passed|failed" $O/gpu_tests.log | tail -12
timeout 900 python tools/recorded_vs_eager_checksums.py --seeds 3 1 --iters 1000 --out $O/recorded_vs_eager.json > $O/recorded_vs_eager.log 2>&1 < /dev/null
tail -3 $O/recorded_vs_eager.log | cut -c1-700
timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2> $O/tsc1024.err < /dev/null | grep '"metric"' > $O/tsc1024.json
python -c "
import json; d=json.load(open('$O/tsc1024.json')); print('tsc1024', round(d['ms_per_step'],2), {k: round(v*1e3,2) for k,v in d.items() if k.endswith('_s') and isinstance(v,float)})"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 6 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/tsc1024_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid > $O/tsc1024_env_step_sequence.txt 2>&1
tail -2 $O/tsc1024_env_step_sequence.txt
