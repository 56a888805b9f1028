#!/bin/bash
# Round 6: the GPU calls as they were run, one case per call (gpurun -- 'bash tools/r6_call.sh <case>').  Outputs under gpurun_out/r6/<case>/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-none}
O=$R/gpurun_out/r6/$C
mkdir -p $O
cd $R
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2), "value", round(d["value"]))
    except Exception as e: print(f, "no line", e)
PY
}
case $C in
base)      # the ADVICE r5 fixes (second gradient of a parameter in parts, learn_vision entered twice), bench.py launching its own ranks, then the whole suite and the default line
    timeout 900 python -m pytest tests/test_grad_parts.py tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log
    timeout 900 python -m pytest tests/test_tsc_depth.py -m gpu -x -q -k "recorded_vision" > $O/pytest_vision.log 2>&1; tail -4 $O/pytest_vision.log
    timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
    timeout 600 python bench.py 2> $O/bench.err < /dev/null | grep '"metric"' > $O/bench_default.json; cut -c1-300 $O/bench_default.json
    QA_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --no_cpu_baseline 2> $O/bench_gpus2.err < /dev/null | grep '"metric"' > $O/bench_gpus2_shared.json; cut -c1-300 $O/bench_gpus2_shared.json
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ;;
lock)      # the learner lockstep (GPU learner vs torch-CPU learner on the same state / rollout / tables), config 3, 1024 envs x 200 iterations;
           # beside it (the lockstep is CPU-bound): the self-collision proximity runs scripted in r5 and never run (item 8), and a fresh per-step error distribution on HEAD's kernel
    timeout 900 python -m pytest tests/test_learner_lockstep.py -m gpu -x -q -s > $O/pytest_lockstep.log 2>&1; tail -4 $O/pytest_lockstep.log
    ( QA_CPU_THREADS=48 timeout 3300 python tools/learner_lockstep.py --amp --num_envs 1024 --iters 200 --out $O/learner_lockstep_cfg3.json > $O/lockstep.log 2>&1 ) &
    ( timeout 1500 python tools/self_collision_proximity.py --amp --num_envs 1024 --iters 600 --out $O/self_collision_proximity_cfg3_1024x600.json > $O/prox_cfg3.log 2>&1
      timeout 2000 python tools/self_collision_proximity.py --tsc --num_envs 1024 --iters 300 --every 5 --out $O/self_collision_proximity_cfg4_1024x300.json > $O/prox_cfg4.log 2>&1 ) &
    timeout 900 python tools/step_error_distribution.py --out $O/step_error_distribution.json > $O/step_error_distribution.log 2>&1; tail -3 $O/step_error_distribution.log | cut -c1-300
    wait
    grep "^it " $O/lockstep.log | tail -12 | cut -c1-700; tail -1 $O/lockstep.log | cut -c1-600
    tail -3 $O/prox_cfg3.log | cut -c1-400; tail -3 $O/prox_cfg4.log | cut -c1-400
    ;;
lock2)     # the lockstep's CONTROL arm (torch-CPU vs torch-CPU from a one-ulp perturbed state) beside the forced arm, 200 iterations; in the foreground the chain case
    ( QA_CPU_THREADS=48 timeout 3000 python tools/learner_lockstep.py --amp --num_envs 1024 --iters 200 --free 0 --control 1e-7 --out $O/learner_lockstep_cfg3_control.json > $O/lockstep.log 2>&1 ) &
    timeout 600 python -m pytest tests/test_learner_lockstep.py -m gpu -x -q -s > $O/pytest_lockstep.log 2>&1; tail -4 $O/pytest_lockstep.log | cut -c1-600
    bash tools/r6_call.sh chain
    wait
    grep "^it " $O/lockstep.log | tail -12 | cut -c1-900; tail -1 $O/lockstep.log | cut -c1-600
    ;;
chain)     # ABI 17: the PPO minibatch step's networks as two chain launches (train_chain.py): parity, then the small-share bench lines with / without
    timeout 900 python -m pytest tests/test_train_chain.py tests/test_policy_chain.py -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -5 $O/pytest_chain.log
    timeout 1500 python -m pytest tests/test_fused_learner.py tests/test_golden_learner.py tests/test_gpu_train.py tests/test_grad_parts.py tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_learner.log 2>&1; tail -5 $O/pytest_learner.log
    for n in 512 1024; do
      timeout 300 python bench.py --num_envs $n --no_cpu_baseline 2>$O/bench_${n}_chain.err < /dev/null | grep '"metric"' > $O/bench_${n}_chain.json
      QA_TRAIN_CHAIN=0 timeout 300 python bench.py --num_envs $n --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_${n}_autograd.json
    done
    timeout 400 python bench.py --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_4096.json
    line $O/bench_*.json
    timeout 200 python tools/policy_time.py 4096 > $O/policy_time_4096.txt 2>&1; tail -2 $O/policy_time_4096.txt
    ;;
stack)     # ABI 18: the discriminator step's optimiser half as one launch; and why the 1024-env task-level line of final_measure read like autograd steps
    timeout 900 python -m pytest tests/test_train_chain.py tests/test_fused_learner.py -m gpu -x -q -k "stacked or discriminator or adam" > $O/pytest_stack.log 2>&1; grep -E "passed|failed" $O/pytest_stack.log | tail -2; grep -E "^FAILED|Error" $O/pytest_stack.log | head -5
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2> $O/bench_cfg3.err < /dev/null | grep '"metric"' > $O/bench_cfg3_stack.json
    QA_DISC_STACKED_ADAM=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_three_optimisers.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512_stack.json
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline > $O/tsc_20_5_nocpu.log 2>&1 < /dev/null; grep '"metric"' $O/tsc_20_5_nocpu.log > $O/bench_tsc1024_20_5_nocpu.json; grep -v '"metric"' $O/tsc_20_5_nocpu.log | grep -i "graph\|capture\|eager\|chain" | head -5
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 > $O/tsc_8_3_cpu.log 2>&1 < /dev/null; grep '"metric"' $O/tsc_8_3_cpu.log > $O/bench_tsc1024_8_3_cpu.json; grep -v '"metric"' $O/tsc_8_3_cpu.log | grep -i "graph\|capture\|eager\|chain" | head -5
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_disc_loss_kernel" mid > $O/disc_step_sequence.txt 2>&1; tail -30 $O/disc_step_sequence.txt
    ;;
stack2)    # 512-element chunks in the stacked optimiser launch; the task-level line with the one-off captures outside the timed region; per-kernel traffic
    timeout 900 python -m pytest tests/test_train_chain.py -m gpu -x -q -k "stacked or discriminator" > $O/pytest_stack.log 2>&1; grep -E "passed|failed" $O/pytest_stack.log | tail -2
    for i in 1 2; do timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2> $O/bench_cfg3.err < /dev/null | grep '"metric"' > $O/bench_cfg3_stack_$i.json; done
    QA_BENCH_TRACE=1 timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2> $O/tsc_trace.err < /dev/null | grep '"metric"' > $O/bench_tsc1024_20_5.json; grep "per-iteration" $O/tsc_trace.err
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_disc_loss_kernel" mid > $O/disc_step_sequence.txt 2>&1; grep -E "stack|launches" $O/disc_step_sequence.txt
    for NE in 1024; do
      timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf$NE -- python $R/tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tf.log 2>&1
      timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw$NE -- python $R/tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tw.log 2>&1
      cd $R; python tools/pmc_to_json.py /tmp/pmc_tf$NE /tmp/pmc_tw$NE $NE $O/tsc_env_step_traffic_$NE.json 0 qa_env_step,qa_tsc_goal_step,qa_tsc_observations
    done
    ;;
stack3)    # several steps per recorded graph (QA_STEP_UNROLL) + the task-level DAgger update as replays
    timeout 2400 python -m pytest tests/test_tsc_learner.py tests/test_train_chain.py tests/test_gpu_train.py tests/test_learner_lockstep.py tests/test_grad_parts.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR|Error" $O/pytest.log | head -5
    for i in 1 2; do timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2> $O/bench_cfg3.err < /dev/null | grep '"metric"' > $O/bench_cfg3_$i.json; done
    QA_STEP_UNROLL=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_one_step_per_replay.json
    for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_$i.json; done
    QA_STEP_UNROLL=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_one_step_per_replay.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    QA_BENCH_TRACE=1 timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2> $O/tsc_trace.err < /dev/null | grep '"metric"' > $O/bench_tsc1024_20_5.json; grep "per-iteration\|capture" $O/tsc_trace.err
    line $O/bench_*.json
    ;;
tests3)    # the suites the stack3 call did not reach
    timeout 2400 python -m pytest tests/test_tsc_learner.py tests/test_train_chain.py tests/test_gpu_train.py tests/test_learner_lockstep.py tests/test_grad_parts.py tests/test_fused_learner.py tests/test_golden_learner.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    ;;
pair)      # ABI 18 second half: two clipped Adam steps + the KL rule as three launches
    timeout 2400 python -m pytest tests/test_tsc_learner.py tests/test_train_chain.py tests/test_gpu_train.py tests/test_learner_lockstep.py tests/test_grad_parts.py tests/test_fused_learner.py tests/test_golden_learner.py tests/test_distributed_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_$i.json; done
    QA_ADAM_PAIR=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_seven_launches.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024.json
    QA_ADAM_PAIR=0 timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_seven_launches.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_step_sequence_512.txt 2>&1; cat $O/ppo_step_sequence_512.txt
    ;;
pair2)     # + qa_pair_losses; why the task-level step did not take the paired optimiser launch
    timeout 2400 python -m pytest tests/test_tsc_learner.py tests/test_train_chain.py tests/test_gpu_train.py tests/test_learner_lockstep.py tests/test_grad_parts.py tests/test_fused_learner.py tests/test_golden_learner.py tests/test_distributed_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_$i.json; done
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_1024.json
    QA_DEBUG_GRAPH=1 timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>$O/tsc.err < /dev/null > $O/tsc.log; grep '"metric"' $O/tsc.log > $O/bench_tsc1024.json; grep -v '"metric"' $O/tsc.log | sort | uniq -c | head -5
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_step_sequence_512.txt 2>&1; cat $O/ppo_step_sequence_512.txt
    ;;
pair3)     # qa_pair_losses through LDS
    timeout 900 python -m pytest tests/test_fused_learner.py tests/test_train_chain.py tests/test_tsc_learner.py -m gpu -q -k "pair or training or recorded" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_$i.json; done
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_step_sequence_512.txt 2>&1; grep -E "pair|launches" $O/ppo_step_sequence_512.txt
    rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_hybrid_ppo_loss_kernel mid > $O/tsc_step_sequence_1024.txt 2>&1; cat $O/tsc_step_sequence_1024.txt
    ;;
pair4)     # the paired launch's tables built before the task-level capture
    timeout 1200 python -m pytest tests/test_tsc_learner.py tests/test_train_chain.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_$i.json; done
    timeout 400 python bench.py --tsc --num_envs 512 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc512.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_hybrid_ppo_loss_kernel mid > $O/tsc_step_sequence_1024.txt 2>&1; tail -1 $O/tsc_step_sequence_1024.txt
    ;;
dbg)       # the task-level recorded update aborts since pair.warm(): where
    QA_DEBUG_GRAPH=1 timeout 300 python bench.py --tsc --num_envs 1024 --steps 2 --warmup 2 --no_cpu_baseline > $O/tsc.log 2>&1 < /dev/null; tail -25 $O/tsc.log | cut -c1-300
    echo ---- without warm
    timeout 300 python bench.py --tsc --num_envs 1024 --steps 2 --warmup 2 --no_cpu_baseline > $O/tsc2.log 2>&1 < /dev/null; tail -5 $O/tsc2.log | cut -c1-300
    echo ---- serialised launches
    AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 QA_TSC_UPDATE_GRAPH=0 timeout 300 python bench.py --tsc --num_envs 1024 --steps 2 --warmup 2 --no_cpu_baseline > $O/tsc3.log 2>&1 < /dev/null; tail -5 $O/tsc3.log | cut -c1-300
    ;;
fin)       # a learner change: its suites and the lines it moves
    timeout 2400 python -m pytest tests/test_fused_learner.py tests/test_train_chain.py tests/test_tsc_learner.py tests/test_gpu_train.py tests/test_grad_parts.py tests/test_golden_learner.py tests/test_learner_lockstep.py tests/test_distributed_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_$i.json; done
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3.json
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024.json
    line $O/bench_*.json
    ;;
prio)      # the discriminator chain's stream at high priority beside the PPO steps' GEMMs; the hybrid sampling launch with its loads up front
    timeout 1200 python -m pytest tests/test_tsc_learner.py tests/test_tsc_glue.py tests/test_tsc_course_env.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for pr in 0 -1 0 -1; do QA_DISC_STREAM_PRIORITY=$pr timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_prio${pr}_$RANDOM.json; done
    timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid | grep -E "hybrid|launches"
    ;;
fastr6)    # config 3, 1024 envs x 1,000 iterations on the round's final learner (chain steps, stacked / paired optimiser launches): the SAME 32 seeds r5's final code ran
    timeout 2500 python tools/d2_many.py --out $O --arms fast:31-62 --workers 4 --job_timeout 900 --budget_s 2000 > $O/d2_many.log 2>&1
    tail -6 $O/d2_many.log
    ;;
tails)     # the last workgroup's serial tails (stacked optimiser's step counters, discriminator tail) spread over its threads
    timeout 2400 python -m pytest tests/test_fused_learner.py tests/test_train_chain.py tests/test_gpu_train.py tests/test_learner_lockstep.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_$i.json; done
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_disc_loss_kernel" mid > $O/disc_step_sequence.txt 2>&1; cat $O/disc_step_sequence.txt
    ;;
cfg2pl)    # config 2 at 4096 envs: the two pair losses of the 24,576-row step as one launch through LDS, A / B on one box
    timeout 1500 python -m pytest tests/test_fused_learner.py tests/test_gpu_train.py tests/test_golden_learner.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do
      timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_one_launch_$i.json
      QA_PAIR_LOSSES=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_five_launches_$i.json
    done
    line $O/bench_*.json
    ;;
epb)       # qa_rollout_act_hybrid with few envs per workgroup
    timeout 1800 python -m pytest tests/test_tsc_learner.py tests/test_tsc_glue.py tests/test_tsc_course_env.py tests/test_tsc_env.py tests/test_tsc_student.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_$i.json; done
    timeout 600 python bench.py --tsc --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc8192.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
    timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid | grep -E "hybrid|launches"
    ;;
store)     # the observation rows' storage copy inside the sampling launch
    timeout 2400 python -m pytest tests/test_fused_learner.py tests/test_gpu_train.py tests/test_golden_learner.py tests/test_learner_lockstep.py tests/test_full_size_properties.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do
      timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_store_$i.json
      QA_ACT_STORE=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_copy_$i.json
    done
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_store.json
    QA_ACT_STORE=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_copy.json
    line $O/bench_*.json
    ;;
pmcchain)  # SQ counters of the PPO training step's launches at 3,072 rows
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_c
    timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc_c -- python $R/tools/pmc_policy.py train 3072 < /dev/null > /tmp/pmc_c.log 2>&1; tail -2 /tmp/pmc_c.log
    cd $R; python tools/pmc_policy.py summarize /tmp/pmc_c train > $O/train_chain_sq_counters.txt 2>&1; cat $O/train_chain_sq_counters.txt
    ;;
tsccurve)  # the task-level teacher's training curves, chain steps against autograd steps: 8 seeds x 300 iterations at 1024 envs, four processes at a time
    for s in 1 2 3 4 5 6 7 8; do
      ( timeout 600 python tools/tsc_train_curve.py --seed $s --iters 300 --num_envs 1024 --out $O/chain_s$s.json > $O/chain_s$s.log 2>&1; tail -1 $O/chain_s$s.log ) &
      ( QA_TRAIN_CHAIN=0 timeout 600 python tools/tsc_train_curve.py --seed $s --iters 300 --num_envs 1024 --out $O/autograd_s$s.json > $O/autograd_s$s.log 2>&1; tail -1 $O/autograd_s$s.log ) &
      if [ $((s % 2)) -eq 0 ]; then wait; fi
    done
    wait
    python tools/tsc_train_curve.py merge $O/merged.json $O/chain_s*.json -- $O/autograd_s*.json
    rm -f $O/*.log
    ;;
tscepoch)  # task-level learner: rollout gathered once per update, the epoch's slots as one recording
    timeout 1800 python -m pytest tests/test_tsc_learner.py tests/test_tsc_glue.py tests/test_train_chain.py tests/test_distributed_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -8
    for i in 1 2; do timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_$i.json; done
    QA_STEP_UNROLL=0 timeout 400 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_step_per_replay.json
    timeout 400 python bench.py --tsc --num_envs 512 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc512.json
    line $O/bench_*.json
    ;;
traffic)   # which of the task-level env step's three launches moves the bytes (per-kernel FETCH_SIZE / WRITE_SIZE)
    cd /tmp && export TMPDIR=/tmp
    for NE in 1024; do
      timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf$NE -- python $R/tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tf.log 2>&1
      timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw$NE -- python $R/tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tw.log 2>&1
      cd $R; python tools/pmc_to_json.py /tmp/pmc_tf$NE /tmp/pmc_tw$NE $NE $O/tsc_env_step_traffic_$NE.json 0 qa_env_step,qa_tsc_goal_step,qa_tsc_observations
    done
    ;;
suite)     # the driver's round-end tiers on HEAD: the whole GPU suite, smoke(), the default bench line
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err < /dev/null | grep '"metric"' > $O/bench_default.json; cut -c1-300 $O/bench_default.json
    ;;
flaky)     # is the 8192-env recorded-vs-eager difference run-to-run noise?  the same test four times, chains on / off
    for i in 1 2; do
      timeout 600 python -m pytest "tests/test_tsc_learner.py::test_recorded_update_equals_eager_update[8192-False]" -m gpu -q 2>&1 | grep -E "differ by|passed|failed" | head -3
      QA_TRAIN_CHAIN=0 timeout 600 python -m pytest "tests/test_tsc_learner.py::test_recorded_update_equals_eager_update[8192-False]" -m gpu -q 2>&1 | grep -E "differ by|passed|failed" | head -3
    done
    ;;
tsc)       # the task-level learner's step as chain launches: parity, its suites, the 1024-env line with / without
    timeout 900 python -m pytest tests/test_train_chain.py -m gpu -x -q > $O/pytest_chain.log 2>&1; grep -E "passed|failed" $O/pytest_chain.log | tail -2
    timeout 1800 python -m pytest tests/test_tsc_learner.py tests/test_tsc_env.py tests/test_tsc_course_env.py tests/test_tsc_glue.py tests/test_tsc_student.py tests/test_tsc_depth.py -m gpu -q > $O/pytest_tsc.log 2>&1; grep -E "passed|failed|^FAILED" $O/pytest_tsc.log | tail -5
    for i in 1 2; do
      timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2>$O/bench_tsc.err < /dev/null | grep '"metric"' > $O/bench_tsc1024_chain_$i.json
    done
    QA_TRAIN_CHAIN=0 timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_autograd.json
    timeout 400 python bench.py --tsc --num_envs 512 --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc512_chain.json
    QA_TRAIN_CHAIN=0 timeout 400 python bench.py --tsc --num_envs 512 --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc512_autograd.json
    line $O/bench_*.json
    ;;
quick)     # a plan change in the batched products: parity + the lines + one step each
    timeout 900 python -m pytest tests/test_train_chain.py tests/test_grad_parts.py tests/test_golden_learner.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
    for i in 1 2; do
      timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_chain_$i.json
      timeout 400 python bench.py --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_chain_$i.json
    done
    timeout 300 python bench.py --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_1024_chain.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_chain_step_sequence_512.txt 2>&1
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_disc_loss_kernel mid > $O/disc_chain_step_sequence_4096.txt 2>&1
    cd $R
    grep -E "wgrad|launches" $O/ppo_chain_step_sequence_512.txt $O/disc_chain_step_sequence_4096.txt | cut -c1-160
    ;;
pad)       # batch + padded-row 16-byte loads + one pack launch + row limit 8192; crossover at 2048 envs
    QA_TRAIN_CHAIN_SIDES=0 bash tools/r6_call.sh sides
    O=$R/gpurun_out/r6/sides
    timeout 300 python bench.py --num_envs 2048 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_2048.json
    QA_TRAIN_CHAIN_MAX_ROWS=16384 timeout 300 python bench.py --num_envs 2048 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_2048_chain.json
    QA_TRAIN_CHAIN=0 timeout 300 python bench.py --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_1024_autograd.json
    QA_DISC_TRAIN_CHAIN=0 timeout 300 python bench.py --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512_ppo_chain_only.json
    line $O/bench_2048*.json $O/bench_1024*.json $O/bench_cfg3_512*.json
    ;;
batch)     # the chain steps' weight-gradient products in one call (qa_linear_backward_weight_batch): parity, suites, lines, traces
    QA_TRAIN_CHAIN_SIDES=0 bash tools/r6_call.sh sides
    ;;
sides)     # chain steps with side streams, loads instead of copies: parity, the learner suites, the lines, one step of each in the trace
    timeout 900 python -m pytest tests/test_train_chain.py tests/test_policy_chain.py -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -3 $O/pytest_chain.log | cut -c1-300
    timeout 1800 python -m pytest tests/test_fused_learner.py tests/test_golden_learner.py tests/test_gpu_train.py tests/test_grad_parts.py tests/test_disc_step_tail.py tests/test_distributed_gpu.py tests/test_hybrid_arm.py tests/test_learner_lockstep.py -m gpu -x -q > $O/pytest_learner.log 2>&1; tail -3 $O/pytest_learner.log | cut -c1-300
    for i in 1 2; do
      timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>$O/bench_512.err < /dev/null | grep '"metric"' > $O/bench_512_chain_$i.json
      timeout 400 python bench.py --amp --no_cpu_baseline 2>$O/bench_amp.err < /dev/null | grep '"metric"' > $O/bench_cfg3_chain_$i.json
    done
    timeout 300 python bench.py --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_1024.json
    QA_TRAIN_CHAIN_MAX_ROWS=8192 timeout 300 python bench.py --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_1024_chain.json
    QA_TRAIN_CHAIN=0 timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_autograd.json
    QA_TRAIN_CHAIN=0 timeout 400 python bench.py --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_autograd.json
    timeout 300 python bench.py --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512_chain.json
    timeout 400 python bench.py --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_4096.json
    line $O/bench_*.json
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_chain_step_sequence_512.txt 2>&1
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_disc_loss_kernel mid > $O/disc_chain_step_sequence_4096.txt 2>&1
    cd $R
    tail -1 $O/ppo_chain_step_sequence_512.txt; tail -1 $O/disc_chain_step_sequence_4096.txt
    ;;
trace)     # what one chain step is made of: kernel traces of the 512-env config-2 line and of config 3, one step each
    timeout 600 python -m pytest tests/test_train_chain.py -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -3 $O/pytest_chain.log | cut -c1-300
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_chain_step_sequence_512.txt 2>&1
    f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $O/ppo_chain_512_kernel_stats_head.csv
    rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_disc_loss_kernel mid > $O/disc_chain_step_sequence_4096.txt 2>&1
    [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_step_sequence_cfg3_4096.txt 2>&1
    cd $R
    cat $O/ppo_chain_step_sequence_512.txt | cut -c1-150; cat $O/disc_chain_step_sequence_4096.txt | cut -c1-150
    ;;
dchain2)   # dchain + the PPO chain's small-share lines on a quiet box (the `chain` lines of call lock2 ran beside a 48-thread CPU job)
    bash tools/r6_call.sh dchain
    for n in 512 1024; do
      timeout 300 python bench.py --num_envs $n --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_${n}_chain.json
      QA_TRAIN_CHAIN=0 timeout 300 python bench.py --num_envs $n --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_${n}_autograd.json
    done
    line $O/bench_*.json
    ;;
dchain)    # the discriminator step's three chain launches (train_chain.DiscTrainChain): parity, the AMP tests, config 3 with / without
    timeout 900 python -m pytest tests/test_train_chain.py -m gpu -x -q -s > $O/pytest_chain.log 2>&1; tail -6 $O/pytest_chain.log | cut -c1-400
    timeout 1800 python -m pytest tests/test_fused_learner.py tests/test_golden_learner.py tests/test_gpu_train.py tests/test_disc_step_tail.py tests/test_distributed_gpu.py tests/test_hybrid_arm.py -m gpu -x -q > $O/pytest_learner.log 2>&1; tail -5 $O/pytest_learner.log | cut -c1-400
    for i in 1 2; do
      timeout 400 python bench.py --amp --no_cpu_baseline 2>$O/bench_amp_chain.err < /dev/null | grep '"metric"' > $O/bench_cfg3_chain_$i.json
      QA_DISC_TRAIN_CHAIN=0 timeout 400 python bench.py --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_autograd_$i.json
    done
    timeout 300 python bench.py --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512_chain.json
    QA_DISC_TRAIN_CHAIN=0 timeout 300 python bench.py --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_512_autograd.json
    # where the PPO step's chain stops paying: 2048 envs (12,288-row minibatches) and the 4096-env headline (24,576 rows) with the row limit lifted
    QA_TRAIN_CHAIN_MAX_ROWS=32768 timeout 300 python bench.py --num_envs 2048 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_2048_chain.json
    timeout 300 python bench.py --num_envs 2048 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_2048_autograd.json
    QA_TRAIN_CHAIN_MAX_ROWS=32768 timeout 400 python bench.py --no_cpu_baseline 2>$O/bench_4096_chain.err < /dev/null | grep '"metric"' > $O/bench_cfg2_4096_chain.json
    timeout 400 python bench.py --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_4096_autograd.json
    line $O/bench_*.json
    ;;
*) echo "unknown case $C"; exit 2;;
esac
ls -la $O
