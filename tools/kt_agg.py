import csv, collections, re, sys
# per (kernel, grid size): launches, total, median, min from a rocprofv3 kernel_trace.csv.   usage: kt_agg.py <csv> [name filter]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")[:90]
    g=r.get("Grid_Size") or r.get("Grid_Size_X") or ""
    agg[(k, g)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
rows=sorted(agg.items(), key=lambda kv:-sum(kv[1]))
flt=sys.argv[2] if len(sys.argv)>2 else None
if flt: rows=[r for r in rows if flt in r[0][0]]
for (k,g),v in (rows if flt else rows[:40]):
    v=sorted(v); print(f"{k:92s} grid {g:>9s} n {len(v):4d} total {sum(v)/1e3:8.2f} ms median {v[len(v)//2]:8.1f} us min {v[0]:8.1f}")
