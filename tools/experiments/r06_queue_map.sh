#!/bin/bash
# round 6: which hardware queue each kernel family runs on (rocprofv3 kernel trace, Queue_Id), config 3 with one and with two solver streams
O=gpurun_out/r6x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ss in 2 1; do
  SPVO_TUNE_SOLVE_STREAMS=$ss rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_$ss -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --no-cpu-baseline --no-extras --no-profile --steps 60 --warmup 30 --repeats 2 > $GRAFT_REPO_ROOT/$O/tr_$ss.json 2> $GRAFT_REPO_ROOT/$O/tr_$ss.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for ss in (2, 1):
    fs = glob.glob("gpurun_out/r6x/trace_%d/**/*kernel_trace.csv" % ss, recursive=True)
    if not fs: print("no trace", ss); continue
    rows = list(csv.DictReader(open(fs[0])))
    print("solver streams", ss, len(rows), "kernels; columns", list(rows[0].keys())[:14])
    q = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    for r in rows[len(rows) // 3:]:
        name = r["Kernel_Name"].split("(")[0][-50:]
        q[name][r.get("Queue_Id")] += 1
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in sorted(q, key=lambda n: -sum(dur[n])):
        print("   %-52s n %5d avg %7.1f us  queues %s" % (name, len(dur[name]), sum(dur[name]) / len(dur[name]), dict(q[name])))
PY
find $O -name "*.csv" -size +20M -delete; find $O -name "*.db" -delete
