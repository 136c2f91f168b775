#!/bin/bash
# Round-4 experiment 6: XCD-banded tile assignment in launches without the tile counters (single-round layers): time and fetched bytes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WINO_DYNAMIC=0
{
for v in "" _nobands; do
  echo "=== wino_bench4$v / wino_bench2n$v"
  $ROOT/tools/wino_bench4$v 90 294 64 128 0 50 240
  $ROOT/tools/wino_bench4$v 90 294 128 128 1 50 240
  $ROOT/tools/wino_bench4$v 45 147 128 512 0 50 240
  $ROOT/tools/wino_bench2n$v 45 147 128 128 0 50 240
done
} > $OUT/sweep6.log 2>&1
for v in "" _nobands; do
  for shape in "90 294 128 128 1 20 240" "45 147 128 512 0 20 240"; do
    tag=$(echo $shape | tr ' ' '_')$v
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$tag -o p -- $ROOT/tools/wino_bench4$v $shape > /dev/null 2>&1
    python3 - $OUT/f_$tag "$shape$v" >> $OUT/sweep6.log <<'PY'
import csv,glob,sys,os
vals=[]
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "conv_wino" in r["Kernel_Name"]: vals.append(float(r["Counter_Value"]))
vals=sorted(vals)[len(vals)//5:]
print(f"FETCH_SIZE {sys.argv[2]}: {sum(vals)/len(vals)/1024:.1f} MB reported (x2 = {2*sum(vals)/len(vals)/1024:.1f} MB fetched)")
PY
  done
done
