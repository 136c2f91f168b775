#!/bin/bash
# HBM traffic of the stand-alone conv1b launch, static against dynamic (XCD-banded) tile assignment: FETCH_SIZE / WRITE_SIZE, separate passes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for d in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pw_$d_$c
    WINO_DYNAMIC=$d rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pw_${d}_$c -o p -- $ROOT/tools/wino_bench2 360 1176 64 64 1 20 238 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pw_${d}_$c/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "conv_wino2" in r["Kernel_Name"] and r["Counter_Name"]=="$c"]
v=sorted(v)[len(v)//5:]
print("dynamic=$d $c: %.1f MB per launch (raw counter, KB -> MB)" % (sum(v)/len(v)/1024))
PY
  done
done
