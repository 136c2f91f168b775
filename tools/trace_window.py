import csv, glob, os, sys
d=sys.argv[1]
rows=[]
for f in glob.glob(os.path.join(d,"**","*kernel_trace.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void spvo::","").replace("spvo::","")[:34],r.get("Queue_Id","")))
rows.sort()
# find a heads kernel in the middle
idx=[i for i,r in enumerate(rows) if r[2].startswith("heads_fused")]
i=idx[len(idx)//2]
t0=rows[i][0]
for r in rows[i-3:i+40]:
    print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} q{r[3]} {r[2]}")
