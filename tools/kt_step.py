import csv, re, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
adam=[i for i,r in enumerate(rows) if "k_adam" in r[2]]
a,b=adam[-4],adam[-3]   # one steady-state step
t0=rows[a][1]
for s,e,k in rows[a+1:b+1]:
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  {k[:110]}")
print("step wall", (rows[b][1]-rows[a][1])/1e3, "us; busy", sum(e-s for s,e,k in rows[a+1:b+1])/1e3)
