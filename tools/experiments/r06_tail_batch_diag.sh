#!/bin/bash
# round 6: config 3 with one tail chain per group against one per pair: the pipeline's own diagnostics (trunk_timing) and a kernel trace of each
O=gpurun_out/r6u; mkdir -p $O
for mode in 1 0; do
  SPVO_TUNE_TAIL_BATCH=$mode SPVO_TUNE_TRUNK_TIMING=1 python bench.py --config 3 --no-cpu-baseline --no-extras --no-profile > $O/b_$mode.json 2> $O/b_$mode.err
  python - $mode <<'PY'
import json, sys
mode = sys.argv[1]
d = json.loads(open("gpurun_out/r6u/b_%s.json" % mode).read().strip().splitlines()[-1])
err = [l.strip() for l in open("gpurun_out/r6u/b_%s.err" % mode) if "[spvo]" in l or "timing]" in l]
print("tail_batch", mode, d["value"], d["ms_per_step_min"], d["ms_per_step_max"], d["latency_ms"]["p50"])
for key in ("trunk timing", "tail stream", "host:", "pairs per launch"):
    for l in [l for l in err if key in l][-1:]: print("    ", l[:330])
PY
done
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  SPVO_TUNE_TAIL_BATCH=$mode rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$mode -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --no-cpu-baseline --no-extras --no-profile --steps 100 --warmup 20 > $GRAFT_REPO_ROOT/$O/tr_$mode.json 2> $GRAFT_REPO_ROOT/$O/tr_$mode.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for mode in (1, 0):
    fs = glob.glob("gpurun_out/r6u/trace_%d/**/*kernel_stats.csv" % mode, recursive=True)
    if not fs: print("no stats", mode); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("tail_batch", mode)
    for r in rows[:22]: print("   %-70s calls %6s avg %8.1f us total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
# keep only the stats, not the traces (64 MiB limit)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
