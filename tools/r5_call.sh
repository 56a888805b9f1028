#!/bin/bash
# Round 5: the GPU calls as they were run, one case per call (gpurun -- 'bash tools/r5_call.sh <case>').  Outputs under gpurun_out/r5/<case>/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-none}
O=$R/gpurun_out/r5/$C
mkdir -p $O
cd $R
case $C in
valu)      # what one wavefront alone can issue (tools/valu_rate.hip) + today's baseline lines
    timeout 120 tools/_prof/valu_rate > $O/valu_rate.txt 2>&1
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_512.json
    timeout 200 python tools/substep_profile.py > $O/substep_profile.txt 2>&1 < /dev/null
    timeout 200 python tools/quick_time.py > $O/quick_time.txt 2>&1 < /dev/null
    ;;
envpk)     # packed rows + sweeps (r5) against the round-4 scalar build (tools/_prof/libqa_sim_scalar.so = -DQA_PGS_SCALAR): parity first, then timing, stamps, SQ counters
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_self_collision.py tests/test_articulated_obstacles.py tests/test_full_size_properties.py tests/test_mocap_reset.py -m gpu -x -q > $O/pytest_env.log 2>&1; tail -5 $O/pytest_env.log
    for i in 1 2; do
      timeout 200 python tools/quick_time.py > $O/quick_time_packed_$i.txt 2>&1 < /dev/null
      QA_LIB=$R/tools/_prof/libqa_sim_scalar.so timeout 200 python tools/quick_time.py > $O/quick_time_scalar_$i.txt 2>&1 < /dev/null
    done
    timeout 300 python tools/substep_profile.py > $O/substep_profile_packed.txt 2>&1 < /dev/null
    timeout 300 python tools/substep_profile.py -DQA_PGS_SCALAR > $O/substep_profile_scalar.txt 2>&1 < /dev/null
    export TMPDIR=/tmp
    SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
    rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/pmc_sq -- python $R/tools/pmc_env_step.py 4096 3 < /dev/null > /tmp/pmc_sq.log 2>&1
    python $R/tools/pmc_tsc_env.py summarize /tmp/pmc_sq > $O/env_step_sq_counters_packed.txt 2>&1
    rm -rf /tmp/pmc_sq2; QA_LIB=$R/tools/_prof/libqa_sim_scalar.so timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/pmc_sq2 -- python $R/tools/pmc_env_step.py 4096 3 < /dev/null > /tmp/pmc_sq2.log 2>&1
    python $R/tools/pmc_tsc_env.py summarize /tmp/pmc_sq2 > $O/env_step_sq_counters_scalar.txt 2>&1
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 200 python tools/ab_lockstep.py $R/tools/_prof/libqa_sim_scalar.so $R/tools/_prof/libqa_sim_packed.so 30 1024 1.0 > $O/lockstep.txt 2>&1; tail -4 $O/lockstep.txt | cut -c1-200
    ;;
# (dbg .. dbg6, removed: the six bisect calls behind profiles/r5_packed_sweeps_debug.txt; they drove debug switches in qa_physics.h that were
#  deleted with the fix -- a hipcc miscompile of a lane-divergent vector-element copy, DESIGN 4.1b.  tools/ab_step.py, tools/ab_lockstep.py and
#  tools/pgs_unit.hip, the tools they ran, are kept.)
learn1)    # gradients left in parts, finished by the optimiser's first pass: tests, then the bench line with and without
    timeout 1500 python -m pytest tests/test_grad_parts.py tests/test_fused_learner.py tests/test_golden_learner.py tests/test_gpu_train.py tests/test_gemm_layers.py -m gpu -x -q > $O/pytest_learner.log 2>&1; tail -5 $O/pytest_learner.log
    for i in 1 2; do
      QA_DEFER_GRAD_FINISH=0 timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_finish_launches_$i.json
      timeout 400 python bench.py --no_cpu_baseline 2> $O/bench.err < /dev/null | grep '"metric"' > $O/bench_cfg2_deferred_$i.json
    done
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/learn1/bench_*.json")):
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round(d["collection_s"] * 1e3, 2), "update", round(d["learn_s"] * 1e3, 2))
    except Exception as e: print(f, "no line", e)
PY
    ;;
phase)     # where the env-step kernel's wavefront waits: s_memtime stamps per phase; the learner tests; the 512-env share with / without deferred finishes
    timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt
    timeout 1500 python -m pytest tests/test_grad_parts.py tests/test_fused_learner.py tests/test_golden_learner.py tests/test_gpu_train.py -m gpu -q > $O/pytest_learner.log 2>&1; tail -5 $O/pytest_learner.log
    QA_DEFER_GRAD_FINISH=0 timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_finish_launches.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_deferred.json
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/phase/bench_*.json")):
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round(d["collection_s"] * 1e3, 2), "update", round(d["learn_s"] * 1e3, 2))
    except Exception as e: print(f, "no line", e)
PY
    ;;
late)      # history-shift stores after the physics: parity, timing, phases
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py tests/test_mocap_reset.py tests/test_golden_env.py -m gpu -x -q > $O/pytest_env.log 2>&1; tail -3 $O/pytest_env.log
    for i in 1 2; do timeout 200 python tools/quick_time.py > $O/quick_time_$i.txt 2>&1 < /dev/null; grep "N=" $O/quick_time_$i.txt; done
    QA_LIB=$R/tools/_prof/libqa_sim_packed.so timeout 200 python tools/quick_time.py 2>&1 < /dev/null | grep "N=" > $O/quick_time_early_store.txt; cat $O/quick_time_early_store.txt
    timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
    ;;
skip)      # exact early-outs for idle non-foot contacts / joint-limit rows: parity, timing, how often the rows exist
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_self_collision.py tests/test_articulated_obstacles.py tests/test_full_size_properties.py tests/test_mocap_reset.py tests/test_golden_env.py -m gpu -x -q > $O/pytest_env.log 2>&1; tail -3 $O/pytest_env.log
    for i in 1 2; do timeout 200 python tools/quick_time.py > $O/quick_time_$i.txt 2>&1 < /dev/null; grep "N=" $O/quick_time_$i.txt; done
    timeout 300 python tools/substep_profile.py -DQA_EXP_COUNT_EXTRA > $O/substep_profile.txt 2>&1; tail -13 $O/substep_profile.txt
    timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    python -c "
import json; d=json.loads(open('$O/bench_cfg2.json').read()); print('bench', round(d['ms_per_step'],2), 'ms rollout', round(d['collection_s']*1e3,2), 'update', round(d['learn_s']*1e3,2), 'kernel us', round(d['roofline']['kernel_ms']*1e3,1), round(d['roofline']['kernel_ms_back_to_back']*1e3,1), 'frac', round(d['roofline']['frac'],4))"
    ;;
skipab)    # the early-outs inside the bench's own scenario (robots under the initial policy), A/B with a build without them; hybrid arm smoke
    for i in 1 2; do
      for v in product noskip scalar; do
        L=""; [ $v != product ] && L=$R/tools/_prof/libqa_sim_$v.so
        QA_LIB=$L timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_${v}_$i.json
        python -c "
import json; d=json.loads(open('$O/bench_${v}_$i.json').read()); print('$v $i', round(d['ms_per_step'],2), 'ms rollout', round(d['collection_s']*1e3,2), 'kernel us in-rollout', round(d['roofline']['kernel_ms']*1e3,1), 'back to back', round(d['roofline']['kernel_ms_back_to_back']*1e3,1))"
      done
    done
    QA_PARITY_NO_LOG=0 timeout 600 python tools/return_curve_parity.py --side hybrid --num_envs 256 --iters 6 --seeds 1 --out $O/hybrid_smoke_cfg2.json > $O/hybrid_smoke_cfg2.log 2>&1; tail -3 $O/hybrid_smoke_cfg2.log
    timeout 900 python tools/return_curve_parity.py --side hybrid --amp --num_envs 256 --iters 6 --seeds 1 --out $O/hybrid_smoke_cfg3.json > $O/hybrid_smoke_cfg3.log 2>&1; tail -3 $O/hybrid_smoke_cfg3.log
    timeout 600 python tools/return_curve_parity.py --side gpu --amp --num_envs 256 --iters 6 --seeds 1 --out $O/gpu_smoke_cfg3.json > $O/gpu_smoke_cfg3.log 2>&1; tail -2 $O/gpu_smoke_cfg3.log
    python - <<'PY'
import json
for f in ("hybrid_smoke_cfg2", "hybrid_smoke_cfg3", "gpu_smoke_cfg3"):
    try:
        d = json.load(open(f"gpurun_out/r5/skipab/{f}.json")); r = d["rows"][0]
        print(f, "env-steps/s", round(r["env_steps_per_s"]), {k: [round(x, 3) for x in v[-3:]] for k, v in r["curves"].items() if k in ("Train/mean_reward", "Train/mean_episode_length")})
    except Exception as e: print(f, "failed", e)
PY
    ;;
local)     # env-local coordinates inside a step: flip shares of every parity protocol (r4: profiles/r4_parity_flip_shares.txt), then the tests as judged
    QA_PARITY_MEASURE=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_articulated_obstacles.py tests/test_tsc_course_env.py tests/test_self_collision.py tests/test_mocap_reset.py -m gpu -s -q 2>&1 | grep "FLIPSHARE\|passed\|failed\|median env-step" > $O/flip_shares.txt; cat $O/flip_shares.txt | cut -c1-400
    timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_articulated_obstacles.py tests/test_tsc_course_env.py tests/test_self_collision.py tests/test_mocap_reset.py tests/test_full_size_properties.py tests/test_tsc_env.py tests/test_tsc_depth.py tests/test_seam1_reference_env.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
    timeout 900 python tools/return_curve_parity.py --side hybrid --amp --num_envs 256 --iters 6 --seeds 1 --out $O/hybrid_smoke_cfg3.json > $O/hybrid_smoke_cfg3.log 2>&1; tail -2 $O/hybrid_smoke_cfg3.log
    python - <<'PY'
import json
for f in ("hybrid_smoke_cfg3",):
    try:
        d = json.load(open(f"gpurun_out/r5/local/{f}.json")); r = d["rows"][0]
        print(f, "env-steps/s", round(r["env_steps_per_s"]), {k: [round(x, 3) for x in v[-3:]] for k, v in r["curves"].items() if k in ("Train/mean_reward", "Train/mean_episode_length")})
    except Exception as e: print(f, "failed", e)
PY
    timeout 200 python tools/quick_time.py --terrain > $O/quick_time_terrain.txt 2>&1; grep "N=" $O/quick_time_terrain.txt
    timeout 200 python tools/quick_time.py > $O/quick_time.txt 2>&1; grep "N=" $O/quick_time.txt
    ;;
local2)    # env-local coordinates in kernel AND oracle: the flip shares again
    QA_PARITY_MEASURE=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_articulated_obstacles.py tests/test_tsc_course_env.py tests/test_self_collision.py tests/test_mocap_reset.py -m gpu -s -q 2>&1 | grep "FLIPSHARE\|passed\|failed" > $O/flip_shares.txt; cat $O/flip_shares.txt | cut -c1-300
    ;;
hybrid)    # config 3, 1024 envs x 1,000 iterations: this round's fast arm and the hybrid arm (oracle physics on the host cores + GPU learner), many seeds side by side
    timeout 3500 python tools/d2_many.py --out $O --arms fast:1-24 hybrid:1-24 --workers 9 --hybrid_threads 28 --job_timeout 1800 --budget_s 2400 > $O/d2_many.log 2>&1
    tail -30 $O/d2_many.log
    ;;
hybrid2)   # second batch: the hybrid seeds the first call did not finish (one job per seed, all at once: the oracle's OpenMP scaling is poor, so
           # many jobs x few threads uses the 256 host cores better than few x many) and fast-arm seeds 16-30 beside them
    ( timeout 3000 python tools/d2_many.py --out $O --arms fast:16-30 --workers 2 --job_timeout 1200 --budget_s 2400 > $O/d2_many_fast.log 2>&1 ) &
    timeout 3100 python tools/d2_many.py --out $O --arms hybrid:6-8,15-24 --workers 13 --hybrid_threads 18 --job_timeout 2900 --budget_s 600 > $O/d2_many.log 2>&1
    wait
    tail -15 $O/d2_many.log; tail -4 $O/d2_many_fast.log
    ;;
grp)       # (ABI 16) policy chain: two strands side by side in a workgroup at >= a tile per CU; env-step variants (tools/build_variant.py): history stores in
           # front of the last substep, split table staging, 16-byte history accesses -- timing of all, parity of the candidates
    timeout 600 python -m pytest tests/test_policy_chain.py -m gpu -q > $O/pytest_policy.log 2>&1; tail -4 $O/pytest_policy.log
    for n in 4096 16384; do
      echo "rows $n two groups: $(timeout 120 python tools/policy_time.py $n 2>&1 | tail -1)" >> $O/policy_time_groups.txt
      echo "rows $n one group:  $(QA_MLP_GROUPS=1 timeout 120 python tools/policy_time.py $n 2>&1 | tail -1)" >> $O/policy_time_groups.txt
    done; cat $O/policy_time_groups.txt
    # help = two helper wavefronts per workgroup (bias forces | contact candidates of every substep, qa_physics.h phys_substep ROLE 1 / 2 / 3)
    for i in 1 2; do
      for v in base lastsplit all3 help helplastsplit helpall3; do
        echo "$v $i: $(QA_LIB=$R/tools/_prof/libqa_sim_$v.so timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
      done
    done
    for v in last split x4; do
      echo "$v 1: $(QA_LIB=$R/tools/_prof/libqa_sim_$v.so timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
    done; cat $O/quick_time_variants.txt
    for v in help helplastsplit helpall3; do
      QA_LIB=$R/tools/_prof/libqa_sim_$v.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py -m gpu -x -q > $O/pytest_env_$v.log 2>&1; echo "$v: $(tail -1 $O/pytest_env_$v.log)"
    done
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json; cut -c1-600 $O/bench_cfg2.json
    QA_MLP_GROUPS=1 timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_one_group.json; cut -c1-300 $O/bench_cfg2_one_group.json
    # learn_vision: the env half of a step recorded per camera phase
    timeout 900 python -m pytest tests/test_tsc_depth.py -m gpu -q -k "vision" > $O/pytest_vision.log 2>&1; tail -15 $O/pytest_vision.log | cut -c1-300
    for ne in 256 512; do
      timeout 400 python bench.py --tsc --vision --num_envs $ne --steps 5 --warmup 3 --no_cpu_baseline 2> $O/bench_student_$ne.err < /dev/null | grep '"metric"' > $O/bench_student_${ne}_recorded.json
    done
    QA_TSC_ROLLOUT_GRAPH=0 timeout 400 python bench.py --tsc --vision --num_envs 256 --steps 5 --warmup 3 --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_student_256_eager.json
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/grp/bench_*.json")):
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms;", {k: round(v * 1e3, 2) for k, v in d.items() if k in ("collection_s", "learn_s")})
    except Exception as e: print(f, "no line", e)
PY
    ;;
fin1)      # helper wavefronts by launch size (default), policy two-group launch off (default): timing lines first, then the whole GPU suite with the config-2
           # hybrid arm (4096 envs x 300 iterations, oracle physics on the host cores + GPU learner; VERDICT r4 item 4, second half) on the host cores beside it
    for i in 1 2; do
      for v in "" helpx4 base; do
        L=""; [ -n "$v" ] && L=$R/tools/_prof/libqa_sim_$v.so
        echo "${v:-product} $i: $(QA_LIB=$L timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
      done
    done; cat $O/quick_time_variants.txt
    QA_LIB=$R/tools/_prof/libqa_sim_helpx4.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py -m gpu -x -q > $O/pytest_env_helpx4.log 2>&1; echo "helpx4: $(tail -1 $O/pytest_env_helpx4.log)"
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    QA_ENV_HELPERS=0 timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_no_helpers.json
    QA_LIB=$R/tools/_prof/libqa_sim_helpx4.so timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_helpx4.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512.json
    QA_ENV_HELPERS=0 timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_no_helpers.json
    timeout 400 python bench.py --amp --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_cfg3_amp.json
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/fin1/bench_*.json")):
    try:
        d = json.loads(open(f).read()); r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2),
              "kernel us", round((r.get("kernel_ms") or 0) * 1e3, 1), "frac", round(r.get("frac") or 0, 4))
    except Exception as e: print(f, "no line", e)
PY
    export TMPDIR=/tmp
    rm -rf /tmp/prof_st; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python $R/bench.py --tsc --vision --num_envs 256 --steps 3 --warmup 2 --no_cpu_baseline < /dev/null > /tmp/prof_st.log 2>&1 )
    f=$(find /tmp/prof_st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/student_256_kernel_stats.csv; grep '"metric"' /tmp/prof_st.log > $O/bench_student_256_under_rocprof.json
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("student 256 envs, 5 iterations under rocprofv3: kernel time total %.1f ms, %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %6.1f ms %7s calls  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], r["Name"][:110]))
PY
    ( timeout 1500 python tools/d2_many.py --out $O/cfg2 --plain --num_envs 4096 --iters 300 --arms fast:1-6 hybrid:1-6 --workers 4 --hybrid_threads 60 --job_timeout 1400 --budget_s 700 > $O/d2_many_cfg2.log 2>&1 ) &
    timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
    wait
    tail -14 $O/d2_many_cfg2.log
    ;;
fin2)      # helper wavefronts, stages 2 and 3: history shift on the bias helper, non-foot rows on the contact helper (product = both; helpnorows = stage 2 only;
           # helpx4rows = product with 16-byte history accesses); parity of the candidates, then the bench lines
    for i in 1 2; do
      for v in "" helpnorows helpx4rows base; do
        L=""; [ -n "$v" ] && L=$R/tools/_prof/libqa_sim_$v.so
        echo "${v:-product} $i: $(QA_LIB=$L timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
      done
    done; cat $O/quick_time_variants.txt
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py tests/test_full_size_properties.py tests/test_gpu_train.py -m gpu -x -q > $O/pytest_env_product.log 2>&1; echo "product: $(tail -1 $O/pytest_env_product.log)"
    QA_LIB=$R/tools/_prof/libqa_sim_helpx4rows.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py -m gpu -x -q > $O/pytest_env_helpx4rows.log 2>&1; echo "helpx4rows: $(tail -1 $O/pytest_env_helpx4rows.log)"
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    QA_LIB=$R/tools/_prof/libqa_sim_helpx4rows.so timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_helpx4rows.json
    QA_ENV_HELPERS=0 timeout 400 python bench.py --no_cpu_baseline 2> /dev/null < /dev/null | grep '"metric"' > $O/bench_cfg2_no_helpers.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512.json
    timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/fin2/bench_*.json")):
    try:
        d = json.loads(open(f).read()); r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2),
              "kernel us", round((r.get("kernel_ms") or 0) * 1e3, 1), "frac", round(r.get("frac") or 0, 4))
    except Exception as e: print(f, "no line", e)
PY
    ;;
fin3)      # helper wavefronts, stage 4: the step's closing scalar stores by the contact helper (product) against the same build without it (helpnoscal)
    for i in 1 2; do
      for v in "" helpnoscal base; do
        L=""; [ -n "$v" ] && L=$R/tools/_prof/libqa_sim_$v.so
        echo "${v:-product} $i: $(QA_LIB=$L timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
      done
    done; cat $O/quick_time_variants.txt
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py tests/test_full_size_properties.py tests/test_gpu_train.py tests/test_hybrid_arm.py tests/test_seam1_reference_env.py -m gpu -x -q > $O/pytest_env_product.log 2>&1; echo "product: $(tail -1 $O/pytest_env_product.log)"
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512.json
    timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/fin3/bench_*.json")):
    try:
        d = json.loads(open(f).read()); r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2),
              "kernel us", round((r.get("kernel_ms") or 0) * 1e3, 1), "frac", round(r.get("frac") or 0, 4))
    except Exception as e: print(f, "no line", e)
PY
    ;;
fin4)      # helper wavefronts, stage 5: the contact helper stages the constant table, the bias helper's history stores go out one substep earlier
    for i in 1 2 3; do
      for v in "" base; do
        L=""; [ -n "$v" ] && L=$R/tools/_prof/libqa_sim_$v.so
        echo "${v:-product} $i: $(QA_LIB=$L timeout 100 python tools/quick_time.py 2>&1 | grep 'N=' | tr '\n' ' ')" >> $O/quick_time_variants.txt
      done
    done; cat $O/quick_time_variants.txt
    timeout 900 python -m pytest tests/test_hip_parity.py tests/test_mocap_reset.py tests/test_golden_env.py tests/test_self_collision.py tests/test_full_size_properties.py tests/test_gpu_train.py tests/test_seam1_reference_env.py -m gpu -x -q > $O/pytest_env_product.log 2>&1; echo "product: $(tail -1 $O/pytest_env_product.log)"
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json
    timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512.json
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/fin4/bench_*.json")):
    try:
        d = json.loads(open(f).read()); r = d.get("roofline") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2),
              "kernel us", round((r.get("kernel_ms") or 0) * 1e3, 1), "frac", round(r.get("frac") or 0, 4))
    except Exception as e: print(f, "no line", e)
PY
    ;;
final)     # the round's measurement pass (tools/final_measure.sh -> gpurun_out/final/), then the whole GPU suite and smoke() on the code as measured
    timeout 3000 bash tools/final_measure.sh > $O/final_measure.log 2>&1; tail -30 $O/final_measure.log
    timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
    ;;
fast3)     # config 3 fast arm on the round's FINAL code (helper wavefronts are active at 1024 envs): eight more seeds, beside the 29 of the earlier builds
    timeout 1700 python tools/d2_many.py --out $O --arms fast:31-38 --workers 3 --job_timeout 1200 --budget_s 1300 > $O/d2_many.log 2>&1
    tail -10 $O/d2_many.log
    ;;
fast4)     # the helper-vs-one-wavefront test, then 24 more config-3 fast-arm seeds on the final code (with fast3's eight: a fast arm of the shipped kernels only)
    timeout 900 python -m pytest tests/test_env_helpers.py -m gpu -s -q > $O/pytest_env_helpers.log 2>&1; grep -E "worst|passed|failed|Error" $O/pytest_env_helpers.log | cut -c1-400
    timeout 2300 python tools/d2_many.py --out $O --arms fast:39-62 --workers 3 --job_timeout 1200 --budget_s 1750 > $O/d2_many.log 2>&1
    tail -6 $O/d2_many.log
    ;;
last)      # the round's last call, on HEAD: the driver's own commands (default bench line, the whole GPU suite, smoke)
    timeout 600 python bench.py 2> $O/bench.err < /dev/null | grep '"metric"' > $O/bench_default.json; cut -c1-400 $O/bench_default.json
    timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ;;
pro)       # history loads through AGPRs issued last, obs-tail ballot, PostIn preload: parity as judged, then timing / phases / the bench line
    timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_articulated_obstacles.py tests/test_tsc_course_env.py tests/test_self_collision.py tests/test_mocap_reset.py tests/test_full_size_properties.py tests/test_tsc_env.py tests/test_hybrid_arm.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
    for i in 1 2; do
      timeout 200 python tools/quick_time.py > $O/quick_time_$i.txt 2>&1; grep "N=" $O/quick_time_$i.txt
      QA_LIB=$R/tools/_prof/libqa_sim_local.so timeout 200 python tools/quick_time.py > $O/quick_time_before_$i.txt 2>&1; grep "N=" $O/quick_time_before_$i.txt     # the build of commit ad4fa6b
    done
    timeout 200 python tools/quick_time.py --terrain > $O/quick_time_terrain.txt 2>&1; grep "N=" $O/quick_time_terrain.txt
    timeout 200 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
    timeout 400 python bench.py --no_cpu_baseline 2> $O/bench_cfg2.err < /dev/null | grep '"metric"' > $O/bench_cfg2.json; cut -c1-700 $O/bench_cfg2.json
    # (ABI 15) few row tiles: the chain's strands on separate workgroups (qa_policy.hip mlp_strands); QA_MLP_STRANDS=1 = unsplit
    timeout 300 python -m pytest tests/test_policy_chain.py -m gpu -q > $O/pytest_policy.log 2>&1; tail -3 $O/pytest_policy.log
    for n in 512 1024 2048 4096; do
      echo "rows $n split:   $(timeout 120 python tools/policy_time.py $n 2>&1 | tail -1)" >> $O/policy_time_strands.txt
      echo "rows $n unsplit: $(QA_MLP_STRANDS=1 timeout 120 python tools/policy_time.py $n 2>&1 | tail -1)" >> $O/policy_time_strands.txt
    done; cat $O/policy_time_strands.txt
    # the 8-GPU share of config 2 is launch-bound in the update (3,072-row minibatches): every dense layer on this build's GEMMs (one launch per
    # layer forward, ELU' in the input-gradient's epilogue) against the default (library products for the wide layers)
    for v in heads all; do
      QA_OWN_LAYERS=$v timeout 300 python bench.py --num_envs 512 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_512_own_$v.json
      QA_OWN_LAYERS=$v timeout 300 python bench.py --num_envs 1024 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_1024_own_$v.json
      QA_OWN_LAYERS=$v timeout 400 python bench.py --tsc --num_envs 1024 --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null < /dev/null | grep '"metric"' > $O/bench_tsc1024_own_$v.json
    done
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/pro/bench_*own*.json")):
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms; rollout", round((d.get("collection_s") or 0) * 1e3, 2), "update", round((d.get("learn_s") or 0) * 1e3, 2))
    except Exception as e: print(f, "no line", e)
PY
    ;;
prox)      # the self-collision pairs the kernel does not model, on configs 3 and 4: smallest gaps over whole training runs
    timeout 1500 python tools/self_collision_proximity.py --amp --num_envs 1024 --iters 600 --out $O/self_collision_proximity_cfg3_1024x600.json > $O/cfg3.log 2>&1; tail -3 $O/cfg3.log
    timeout 2400 python tools/self_collision_proximity.py --tsc --num_envs 1024 --iters 300 --every 5 --out $O/self_collision_proximity_cfg4_1024x300.json > $O/cfg4.log 2>&1; tail -3 $O/cfg4.log
    ;;
*) echo "unknown case $C"; exit 2;;
esac
ls -la $O
