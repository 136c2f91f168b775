"""Idle time of the network stream inside the pipelined loop: from a rocprofv3 --kernel-trace CSV of `bench.py --no-cpu-baseline --no-extras`,
the gaps between consecutive kernels on the queue that carries the convolutions, and for every gap above a threshold what the other
queues were doing inside it (the last kernel that ended before the network resumed).  usage: pipeline_gaps.py trace_dir [min_gap_us = 20]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void spvo::", "").replace("spvo::", "")[:40], r.get("Queue_Id", "")))
rows.sort()
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
netq = max(byq, key=lambda q: sum(e - s for s, e, n, _ in byq[q] if n.startswith("conv_")))
net = byq[netq]
# steady state: the middle 60 % of the network queue's kernels
lo, hi = int(len(net) * 0.2), int(len(net) * 0.8)
span = (net[hi][0] - net[lo][0]) / 1e3
gaps = []
for a, b in zip(net[lo:hi], net[lo + 1:hi + 1]):
    g = (b[0] - a[1]) / 1e3
    if g > thr:
        others = [r for r in rows if r[3] != netq and a[1] <= r[1] <= b[0]]
        last = max(others, key=lambda r: r[1]) if others else None
        gaps.append((g, a[2], b[2], last[2] if last else "-", (b[0] - last[1]) / 1e3 if last else 0.0, len(others)))
tot = sum(g[0] for g in gaps)
n_pre = sum(1 for r in net[lo:hi] if r[2].startswith("preprocess_kernel"))
print(f"network queue {netq}: {hi - lo} kernels over {span:.0f} us, {n_pre} pairs; {len(gaps)} gaps above {thr:.0f} us = {tot:.0f} us = {tot / max(n_pre, 1):.1f} us per pair")
hist = collections.Counter((g[1], g[2], g[3]) for g in gaps)
for (a, b, last), n in hist.most_common(12):
    gs = [g for g in gaps if (g[1], g[2], g[3]) == (a, b, last)]
    print(f"  {n:4d} x  after {a:28s} before {b:28s}: mean {sum(g[0] for g in gs) / n:6.1f} us; last kernel elsewhere {last:28s} ended {sum(g[4] for g in gs) / n:6.1f} us before the network resumed")
