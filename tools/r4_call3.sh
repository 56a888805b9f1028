#!/bin/bash
# r4 GPU call 3: flip shares under the derived tolerances, the GPU suite, recorded-vs-eager checksums, self-collision proximity
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
QA_PARITY_MEASURE=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_self_collision.py tests/test_articulated_obstacles.py tests/test_tsc_course_env.py -m gpu -q -s 2>&1 < /dev/null | grep -E "FLIPSHARE|passed|failed" > $O/flip_shares.txt
cat $O/flip_shares.txt
timeout 300 python tools/step_error_distribution.py --out $O/step_error_distribution.json > $O/sed.log 2>&1 < /dev/null
python -c "import json; d=json.load(open('$O/step_error_distribution.json')); print('outside', d['outside_tolerance']['union_share_of_env_steps'], d['outside_tolerance']['per_tensor'])"
timeout 1500 python tools/recorded_vs_eager_checksums.py --seeds 3 6 1 --iters 1000 --out $O/recorded_vs_eager.json > $O/recorded_vs_eager.log 2>&1 < /dev/null
tail -4 $O/recorded_vs_eager.log | cut -c1-600
timeout 600 python tools/self_collision_proximity.py --num_envs 1024 --iters 600 --every 10 --out $O/self_collision_proximity_cfg2.json > $O/scp.log 2>&1 < /dev/null
tail -3 $O/scp.log
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 < /dev/null; echo "pytest rc $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
