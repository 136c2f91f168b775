#!/bin/bash
# which hardware queues the context's streams land on (rocprofv3 kernel trace: Queue_Id per kernel family), process by process, beside the frame rate
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5q2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf $O/tr
  rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $ROOT/bench.py --config 3 --no-cpu-baseline --no-extras --no-profile --steps 200 --repeats 3 > $O/bench_$i.log 2>&1
  python3 - $i <<'PY'
import csv, glob, os, sys, json, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5q2"
f = glob.glob(root + "/tr/**/*kernel_trace.csv", recursive=True)[0]
q = collections.defaultdict(collections.Counter)
rows = list(csv.DictReader(open(f)))
for r in rows[len(rows) // 2:]:
    n = r["Kernel_Name"]
    fam = "trunk" if "conv_f16" in n or "conv_first" in n else "heads" if "heads_fused" in n else "tail" if ("nms_" in n or "heatmap" in n or "sample_desc" in n or "match_" in n) else "solve" if ("ransac" in n or "solve_" in n) else None
    if fam: q[fam][r["Queue_Id"]] += 1
val = None
for line in open(root + "/bench_%s.log" % sys.argv[1]):
    if line.startswith('{"metric"'): val = json.loads(line)["value"]
print("run", sys.argv[1], "frames/s under the tracer", val, {k: dict(v) for k, v in q.items()})
PY
done
