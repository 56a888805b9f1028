#!/bin/bash
# r4: rebuild the config-3 return-curve parity profiles from the per-seed run files under gpurun_out/ (GPU calls 1-9 of the round + the local CPU seeds)
cd /root/repo
FAST=""; for s in $(seq 1 20); do FAST="$FAST gpurun_out/d2r4/fast_s$s.json"; done; for s in $(seq 21 40); do FAST="$FAST gpurun_out/d2r4b/fast_s$s.json"; done
CPU="gpurun_out/d2/cpu_cfg3_s1.json gpurun_out/d2/cpu_cfg3_s2.json gpurun_out/d2/cpu_cfg3_s3.json gpurun_out/d2/cpu_cfg3_s4.json $(ls gpurun_out/d2r4/cpu_local/cpu_s*.json 2>/dev/null | sort -V | tr '\n' ' ')"
N=$(echo $CPU | wc -w)
python tools/merge_d2.py profiles/r4_return_curve_parity_cfg3_amp_1024x1000.json "BASELINE config 3 (AMP, real clips), 1024 envs x 1,000 iterations: HIP fast path (env kernel + GPU learner, recorded steps; 40 seeds) vs CPU oracle + CPU torch learner ($N seeds: 1-4 from r3 -- seed 1 on the oracle from before the self-collision rows --, 5+ run in r4 in the build container, ~2 h each on 6 threads)" $FAST -- $CPU
