#!/bin/bash
# round 6: config 3 under the kernel trace -- per-queue busy fractions and a one-millisecond timeline of every kernel (what paces the loop now that
# the solver chain is off the host's critical path)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r6d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $ROOT/bench.py --config ${1:-3} --no-cpu-baseline --no-extras --no-profile --steps 200 --repeats 3 > $O/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r6d"
f = glob.glob(root + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void spvo::", "").replace("spvo::", "")[:40], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[len(rows) * 5 // 8: len(rows) * 7 // 8]
span = (rows[-1][1] - rows[0][0]) / 1e3
out = []
q = collections.defaultdict(float); qn = collections.Counter()
for r in rows: q[r[3]] += (r[1] - r[0]) / 1e3; qn[r[3]] += 1
n_frames = sum(1 for r in rows if r[2].startswith("solve_tail"))
out.append("window %.0f us, %d frames (solve_tail launches) = %.1f us per frame" % (span, n_frames, span / max(n_frames, 1)))
for k in sorted(q): out.append("  queue %s: %5d kernels, busy %.3f of the window (%.1f us per frame)" % (k, qn[k], q[k] / span, q[k] / max(n_frames, 1)))
i0 = next(i for i, r in enumerate(rows) if r[2].startswith("conv_first") and i > len(rows) // 2)
t0 = rows[i0][0]
out.append("timeline (us from a first-layer launch: start, end, queue, kernel):")
for r in rows[i0:]:
    if (r[0] - t0) / 1e3 > 1000: break
    out.append("%9.1f %9.1f q%s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2]))
open(root + "/timeline.log", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
find $O/tr -name "*.csv" -size +3M -delete
tail -c 200 $O/bench.log
