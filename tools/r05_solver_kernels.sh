#!/bin/bash
# the solver's kernels one by one (staged entry points of tests/test_gpu_odometry.py under the kernel trace)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 -m pytest $ROOT/tests/test_gpu_odometry.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5t"
f = glob.glob(root + "/tr/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)): print(r["Name"][:70].ljust(72), r["Calls"].rjust(5), "%9.1f us avg  min %8.1f  max %8.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -2 $O/pytest.log
