"""Reads a rocprofv3 --kernel-trace CSV and prints, for one steady-state forward pass, every dispatch of the network stream with its
duration and the idle time since the previous dispatch ended (kernel boundaries: launch latency + cache write-back)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by_q = collections.defaultdict(list)
for r in rows:
    by_q[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append(r)
# the queue with the largest total kernel time is the network stream
q = max(by_q, key=lambda k: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in by_q[k]))
rs = by_q[q]
names = [r["Kernel_Name"] for r in rs]
# find the period: index distance between consecutive preprocess kernels, take a late one
idx = [i for i, n in enumerate(names) if n.startswith("spvo::preprocess_kernel") or "preprocess_kernel" in n]
if len(idx) < 4:
    print("no preprocess kernels found on queue", q); sys.exit(1)
i0, i1 = idx[-3], idx[-2]
tot_k = tot_g = 0
prev_end = None
for r in rs[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-90s dur %8.2f us   gap before %7.2f us" % (r["Kernel_Name"][:90], (e - s) / 1e3, gap))
    tot_k += (e - s) / 1e3; tot_g += gap
    prev_end = e
print("sum of kernels %.1f us, sum of gaps %.1f us, period %.1f us" % (tot_k, tot_g, (int(rs[i1]["Start_Timestamp"]) - int(rs[i0]["Start_Timestamp"])) / 1e3))
