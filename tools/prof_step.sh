#!/bin/bash
# rocprofv3 kernel trace of the default bench -> kernel stats + the kernel sequence of one PPO minibatch step.  usage: tools/prof_step.sh <out-prefix> [env assignments...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
mkdir -p $(dirname $R/$OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no_cpu_baseline --steps 10 --warmup 4 < /dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/${OUT}_kernel_stats.csv
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/tools/step_sequence.py "$f" > $R/${OUT}_step_sequence.txt 2>&1
grep '"metric"' /tmp/prof.log > $R/${OUT}_bench.json
