#!/bin/bash
# SQ counter pass of the learner GEMM kernel (128 x 128 tile, 24576 x 672 x 512 forward product) -> gpurun_out/r3/gemm_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_g
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_g -- $R/tools/gemm_probe.bin pmc > /tmp/pmc_g.log 2>&1
python3 - <<'P' > $R/gpurun_out/r3/gemm_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qa_gemm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
P
tail -3 /tmp/pmc_g.log >> $R/gpurun_out/r3/gemm_pmc.txt
