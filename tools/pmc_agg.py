import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:60]
    if sys.argv[2:] and sys.argv[2] not in k: continue
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key=(row["Dispatch_Id"]);
    if key not in seen: seen.add(key); cnt[k]+=1
for k, c in agg.items():
    n = cnt[k]
    print(k, "dispatches", n)
    for name, v in sorted(c.items()): print(f"   {name:32s} {v/n:14.0f}")
