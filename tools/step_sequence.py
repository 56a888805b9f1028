"""Dev tool: the kernel sequence of ONE PPO minibatch step (between two qa_ppo_loss launches late in the run), with
durations and the gap before each kernel.  usage: step_sequence.py <kernel_trace.csv> [marker kernel]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
MARK = sys.argv[2] if len(sys.argv) > 2 else "qa_ppo_loss_kernel"        # "qa_disc_loss_kernel": one discriminator step
idx = [i for i, e in enumerate(ev) if MARK in e[2]]
# "mid" (argv[3]): a marker pair from the middle of the trace (the bench's tail times single kernels back to back: no step lives there)
a, b = (idx[len(idx) // 2], idx[len(idx) // 2 + 1]) if len(sys.argv) > 3 and sys.argv[3] == "mid" else (idx[-6], idx[-5])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void at::native::", "", n)
    m = re.match(r"Cijk_(\w+?)_(S_B_Bias|SB)\w*?_(MT\d+x\d+x\d+)", n)
    if m: return f"GEMM {m.group(1)} {m.group(3)}" + (" +bias" if "Bias" in m.group(2) else "")
    return n[:95]
tot = busy = 0
prev_end = ev[a - 1][1]
for s, e, n in ev[a:b]:
    gap = max(0, s - prev_end); prev_end = max(prev_end, e)
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  {short(n)}")
    busy += e - s
print(f"{b - a} launches, busy {busy / 1e3:.1f} us, span {(ev[b][0] - ev[a][0]) / 1e3:.1f} us")
