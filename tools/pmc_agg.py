import csv, sys, collections, re
# per (kernel, grid size): mean counter values per dispatch from a rocprofv3 counter_collection.csv.
# k_scatter_emit serves the main field and the proposal networks with (sometimes) the same grid size: its key also names
# the kernel dispatched before it (k_reduce_dw -> the main field's call, k_prop_reduce -> a proposal network's).
# usage: pmc_agg.py <csv> [name filter]
rows = list(csv.DictReader(open(sys.argv[1])))
disp = {}
for row in rows:
    disp.setdefault(int(row["Dispatch_Id"]), re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:60])
order = sorted(disp)
prev = {d: (disp[order[i - 1]] if i else "") for i, d in enumerate(order)}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for row in rows:
    k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:60]
    if sys.argv[2:] and sys.argv[2] not in k: continue
    g = row.get("Grid_Size") or row.get("Grid_Size_X") or ""
    d = int(row["Dispatch_Id"])
    if "k_scatter_emit" in k:
        k = f"{k} after {prev[d].replace('fnr::', '')[:24]}"
    k = f"{k} grid {g}"
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if d not in seen: seen.add(d); cnt[k] += 1
for k, c in agg.items():
    n = cnt[k]
    print(k, "dispatches", n)
    for name, v in sorted(c.items()): print(f"   {name:32s} {v/n:14.0f}")
