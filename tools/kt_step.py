"""Two consecutive steady-state training steps (one trains the proposal networks, one does not) from a rocprofv3 kernel
trace: the kernels between the k_train_losses launches (one per step) number 38 and 40, with start offsets, durations
and the hardware queue (two queues = the two HIP streams of training.OVERLAP_PROPOSAL_BACKWARD).
usage: kt_step.py <kernel_trace.csv> [first step, default 38]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "")))
rows.sort()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 38
marks = [i for i, r in enumerate(rows) if "k_train_losses" in r[2]]
a, b = marks[first], marks[first + 2]
t0 = rows[a][0]
queues = {}
for s, e, k, q in rows[a:b]:
    queues.setdefault(q, len(queues))
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  q{queues[q]}  {k[:110]}")
wall = (rows[b][0] - rows[a][0]) / 1e3
print(f"2 steps: wall {wall:.1f} us ({wall / 2:.1f} per step); sum of kernel durations {sum(e - s for s, e, k, q in rows[a:b]) / 1e3:.1f}")
