"""One steady-state training step from a rocprofv3 kernel trace: the kernels between two consecutive k_sample_pixels
launches (one per step), with start offsets and durations."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_train_prologue" in r[2] or "k_sample_pixels" in r[2]]
a, b = marks[-4], marks[-3]
t0 = rows[a][0]
for s, e, k in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {k[:110]}")
print("step wall", (rows[b][0] - rows[a][0]) / 1e3, "us; busy", sum(e - s for s, e, k in rows[a:b]) / 1e3)
