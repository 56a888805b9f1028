#!/bin/bash
# Round-end measurement pass on the GPU box (every command under its own timeout, stdin closed): bench lines of the BASELINE configs,
# rocprofv3 kernel stats of the default bench command, PMC FETCH_SIZE / WRITE_SIZE passes of the env-step kernel (separate passes,
# kernel-trace only).  Outputs land in gpurun_out/final/ (small CSV/JSON only); copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
# PMC first: bench.py reads profiles/env_step_traffic.json for roofline.traffic
# (r4) lean = 3: the kernel a training run without AMP launches (qa_set_lean_exports); lean = 0: the reference's exports, for comparison
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python tools/pmc_env_step.py 4096 3 < /dev/null > /tmp/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python tools/pmc_env_step.py 4096 3 < /dev/null > /tmp/pmc_w.log 2>&1
python tools/pmc_to_json.py /tmp/pmc_f /tmp/pmc_w 4096 $O/env_step_traffic.json 3 > $O/pmc.txt 2>&1 && cp $O/env_step_traffic.json profiles/env_step_traffic.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f0 -- python tools/pmc_env_step.py 4096 0 < /dev/null > /tmp/pmc_f0.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w0 -- python tools/pmc_env_step.py 4096 0 < /dev/null > /tmp/pmc_w0.log 2>&1
python tools/pmc_to_json.py /tmp/pmc_f0 /tmp/pmc_w0 4096 $O/env_step_traffic_full_exports.json 0 >> $O/pmc.txt 2>&1
# (r5) the task-level env step (physics + goal step + observations, the three launches of bench.py --tsc's roofline object) at the two env counts its bench lines use
for NE in 8192 1024; do
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf$NE -- python tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tf.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw$NE -- python tools/pmc_tsc_env.py $NE full < /dev/null > /tmp/pmc_tw.log 2>&1
  python tools/pmc_to_json.py /tmp/pmc_tf$NE /tmp/pmc_tw$NE $NE $O/tsc_env_step_traffic_$NE.json 0 qa_env_step,qa_tsc_goal_step,qa_tsc_observations >> $O/pmc.txt 2>&1
done
for NE in 8192 1024; do [ -f $O/tsc_env_step_traffic_$NE.json ] && cp $O/tsc_env_step_traffic_$NE.json profiles/tsc_env_step_traffic_$NE.json; done
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/pmc_sq -- python tools/pmc_env_step.py 4096 3 < /dev/null > /tmp/pmc_sq.log 2>&1
python tools/pmc_tsc_env.py summarize /tmp/pmc_sq > $O/env_step_sq_counters.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json        # the driver's own command (BENCH_rNN.json `cmd`)
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2> $O/bench_cfg3.err < /dev/null | grep '"metric"' > $O/bench_cfg3_amp.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --terrain trimesh --no_cpu_baseline 2> $O/bench_trimesh.err < /dev/null | grep '"metric"' > $O/bench_cfg2_trimesh.json
( export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 QA_FORCE_DATA_PARALLEL=1; timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_dp.err < /dev/null | grep '"metric"' > $O/bench_cfg2_dp_path_1gpu.json )
for NE in 2048 1024 512; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs $NE --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_${NE}_per_gpu.json; done
QA_TRAIN_CHAIN=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_per_gpu_autograd_steps.json
QA_TRAIN_CHAIN=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_amp_autograd_steps.json
# (r6) the same lines with ABI 17's chain launches but without ABI 18's tails (stacked / paired optimiser launches, one launch for the pair losses, several steps per replay)
A17="QA_DISC_STACKED_ADAM=0 QA_ADAM_PAIR=0 QA_PAIR_LOSSES=0 QA_STEP_UNROLL=0 QA_DISC_LOSS_LOGITS=0"
env $A17 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512_per_gpu_abi17_steps.json
env $A17 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_amp_abi17_steps.json
env $A17 timeout 500 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc_teacher_1024_abi17_steps.json
QA_TRAIN_CHAIN=0 timeout 500 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc_teacher_1024_autograd_steps.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --amp --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_amp_512_per_gpu.json
# (r4) the data-parallel code path with TWO ranks sharing this one GPU (gloo through host memory): a bound on the path's own cost, not a scaling measurement
QA_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --scaling strong --no_cpu_baseline 2> $O/bench_shared_strong.err < /dev/null | grep '"metric"' > $O/bench_cfg2_two_ranks_one_gpu_strong.json
QA_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --scaling weak --num_envs 4096 --no_cpu_baseline 2> $O/bench_shared_weak.err < /dev/null | grep '"metric"' > $O/bench_cfg2_two_ranks_one_gpu_weak.json
timeout 600 python bench.py --tsc --steps 20 --warmup 5 2> $O/bench_tsc.err < /dev/null | grep '"metric"' > $O/bench_tsc_teacher_8192.json
timeout 500 python bench.py --tsc --num_envs 1024 --steps 20 --warmup 5 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc_teacher_1024.json
timeout 600 python bench.py --tsc --vision --num_envs 512 --steps 20 --warmup 5 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc_student_512.json
timeout 600 python bench.py --tsc --vision --num_envs 256 --steps 20 --warmup 5 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc_student_256.json
timeout 200 python tools/quick_time.py > $O/quick_time.txt 2>&1 < /dev/null
timeout 200 python tools/quick_time.py --terrain >> $O/quick_time.txt 2>&1 < /dev/null
timeout 200 python tools/policy_time.py 4096 > $O/policy_time.txt 2>&1 < /dev/null
timeout 200 python tools/substep_profile.py > $O/substep_profile.txt 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/gap_report.py "$f" 0.7 > $O/gap_report.txt 2>&1; [ -n "$f" ] && python $R/tools/step_sequence.py "$f" > $O/step_sequence.txt 2>&1
[ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid > $O/rollout_step_sequence.txt 2>&1      # (r4) every launch of ONE env step of the recorded rollout
grep '"metric"' /tmp/prof.log > $O/bench_under_rocprof.json
rm -rf /tmp/prof
# (r6) one chain step each: the PPO step at the 512-env share, the discriminator step of config 3, the task-level step at 1024 envs
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --num_envs 512 --steps 4 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_ppo_loss_kernel mid > $O/ppo_chain_step_sequence_512.txt 2>&1
rm -rf /tmp/prof
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --amp --steps 3 --warmup 3 --no_cpu_baseline < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_disc_loss_kernel mid > $O/disc_chain_step_sequence_4096.txt 2>&1
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --tsc --num_envs 1024 --steps 6 --warmup 3 < /dev/null > /tmp/prof2.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_tsc_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" "qa_env_step_kernel" mid > $O/tsc_env_step_sequence.txt 2>&1
[ -n "$f" ] && python $R/tools/step_sequence.py "$f" qa_hybrid_ppo_loss_kernel mid > $O/tsc_chain_step_sequence_1024.txt 2>&1
rm -rf /tmp/prof
ls -la $O
