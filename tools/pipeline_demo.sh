#!/bin/bash
# End-to-end pipeline on one MI355X with the reference's command lines: behaviour controller (AMP) -> task-level teacher with that
# frozen controller -> depth student distilled from the teacher.  Writes the scalar logs' last lines to gpurun_out/pipe/summary.txt.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/pipe; rm -rf $O; mkdir -p $O
BBC_ITERS=${BBC_ITERS:-3000}; TEA_ITERS=${TEA_ITERS:-1500}; STU_ITERS=${STU_ITERS:-300}
t0=$(date +%s)
timeout 900 python -m quadrupedal_agility_amd.legged_gym.scripts.train --task go2_locomotion --terrain plane --num_envs 4096 --max_iterations $BBC_ITERS --log_root $O/bbc > $O/bbc.log 2>&1 < /dev/null
t1=$(date +%s)
BBC=$(ls $O/bbc/*/model.pt | head -1)
timeout 900 python -m quadrupedal_agility_amd.tsc.legged_gym.scripts.train --task go2 --headless --num_envs 4096 --randomize_base_mass --randomize_base_com --push_robots --randomize_start --max_iterations $TEA_ITERS --bbc_path $BBC --log_root $O/teacher --exptid teacher > $O/teacher.log 2>&1 < /dev/null
t2=$(date +%s)
mkdir -p quadrupedal_agility_amd/logs/agility; rm -rf quadrupedal_agility_amd/logs/agility/teacher; cp -r $O/teacher quadrupedal_agility_amd/logs/agility/teacher
timeout 900 python -m quadrupedal_agility_amd.tsc.legged_gym.scripts.train --task go2 --headless --use_camera --resume --resumeid teacher --randomize_start --max_iterations $STU_ITERS --bbc_path $BBC --log_root $O/student --exptid student > $O/student.log 2>&1 < /dev/null
t3=$(date +%s)
python - <<PY > $O/summary.txt
import json, glob
def curve(path, tags):
    rows = [json.loads(l) for l in open(path)]
    out = {}
    for t in tags:
        v = [r["value"] for r in rows if r["tag"] == t]
        if v: out[t] = {"first10": sum(v[:10]) / len(v[:10]), "last10": sum(v[-10:]) / len(v[-10:]), "n": len(v)}
    return out
res = {"seconds": {"bbc": $t1 - $t0, "teacher": $t2 - $t1, "student": $t3 - $t2}, "iterations": {"bbc": $BBC_ITERS, "teacher": $TEA_ITERS, "student": $STU_ITERS}}
for name, pat, tags in (("bbc", "$O/bbc/*/scalars.jsonl", ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_reward_i", "Train/mean_episode_length"]),
                        ("teacher", "$O/teacher/scalars.jsonl", ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_episode_length", "Train/success_rate", "Episode/rew_reach_goal"]),
                        ("student", "$O/student/scalars.jsonl", ["Loss_depth/depth_actor", "Loss_depth/yaw", "Loss_depth/obst_type", "Loss_depth/byol", "Train/mean_reward", "Train/success_rate"])):
    f = glob.glob(pat)
    res[name] = curve(f[0], tags) if f else "no log"
print(json.dumps(res, indent=1))
PY
for f in $O/bbc.log $O/teacher.log $O/student.log; do tail -n 2 $f | cut -c1-300; done
find $O -name "*.pt" -delete; rm -rf quadrupedal_agility_amd/logs/agility/teacher      # checkpoints are large: only the scalar logs and the summary travel back
cat $O/summary.txt
