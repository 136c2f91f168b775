#!/bin/bash
# config 3 under the kernel trace: the solver chain's kernels, their durations and the gaps between them inside the pipelined loop
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $ROOT/bench.py --config ${1:-3} --no-cpu-baseline --no-extras --no-profile --steps 200 --repeats 3 > $O/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5p"
f = glob.glob(root + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[len(rows) // 2:]          # the timed half
names = ("solve_in_triangulate", "ransac_hypothesis", "solve_tail", "ransac_select", "solve_gate", "pnp_refine")
sol = [r for r in rows if any(n in r[2] for n in names)]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); chain = []
for i, r in enumerate(sol):
    k = next(n for n in names if n in r[2]); dur[k].append((r[1] - r[0]) / 1e3)
    if i and "solve_in_triangulate" not in r[2]: gap[k].append((r[0] - sol[i - 1][1]) / 1e3)
    if "solve_in_triangulate" in r[2]: t0 = r[0]
    if ("solve_tail" in r[2] or "pnp_refine" in r[2]) and i: chain.append((r[1] - t0) / 1e3)
out = []
for k in names:
    if dur[k]: out.append("%-22s n=%5d  duration mean %7.1f us  median %7.1f   gap in front mean %6.1f us" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]), sorted(dur[k])[len(dur[k]) // 2], (sum(gap[k]) / len(gap[k])) if gap[k] else 0))
chain.sort()
out.append("chain (first kernel's start -> last kernel's end): mean %.1f us, median %.1f, p90 %.1f (n=%d)" % (sum(chain) / len(chain), chain[len(chain) // 2], chain[int(len(chain) * 0.9)], len(chain)))
allk = collections.defaultdict(lambda: [0, 0.0])
for r in rows: allk[r[2][:60]][0] += 1; allk[r[2][:60]][1] += (r[1] - r[0]) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
out.append("window %.0f us; busiest kernels (sum of durations / window):" % span)
for k, (n, t) in sorted(allk.items(), key=lambda kv: -kv[1][1])[:14]: out.append("  %-60s n=%6d  %6.1f us each  %5.3f" % (k, n, t / n, t / span))
open(root + "/solve_chain.log", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
find $O/tr -name "*.csv" -size +3M -delete
tail -c 300 $O/bench.log
