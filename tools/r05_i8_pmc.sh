#!/bin/bash
# SQ counters of the INT8 engine's kernels (tools/perop_int8.py: forward-only loop)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/perop_int8.py mbv1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $ROOT/tools/perop_int8.py mbv1 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5d"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in acc.items():
        if "dwpw" not in k and "conv_i8" not in k: continue
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write("   %-26s mean %.4g (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read()[:6000])
PY
find $OUT -name "*.csv" -size +2M -delete
