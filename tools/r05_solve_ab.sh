#!/bin/bash
# Same-box A/B of the working tree against a SIDE BUILD of another revision.  The side build is made here, before the GPU call:
#   git stash (or git worktree of the other revision); make -C superpoint-stereo-visual-odometry_amd -j8 BUILD=build_base OUT=variants/base; git stash pop; make -C ... -j8
# (variants/ is git-ignored and travels with the snapshot; spvo/capi.py and spvo/host.py load from SPVO_LIB_DIR when it is set)
# the fused solve's chain shortened: parity tests, then config 3 / 5, the host legs and the headline against the side build variants/base
O=gpurun_out/r5s; mkdir -p $O
python -m pytest tests/test_gpu_odometry.py tests/test_gpu_host.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
V=$PWD/superpoint-stereo-visual-odometry_amd/variants/base
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export SPVO_LIB_DIR=$V; else unset SPVO_LIB_DIR; fi
    python bench.py --config 3 --no-cpu-baseline --no-extras > $O/cfg3_${lib}_$rep.json 2> /dev/null
    python bench.py --config 5 --no-cpu-baseline --no-extras > $O/cfg5_${lib}_$rep.json 2> /dev/null
    python bench.py --no-cpu-baseline --legs host > $O/head_${lib}_$rep.json 2> /dev/null
  done
done
unset SPVO_LIB_DIR
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5s/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    hi = d.get("host_interface") or {}
    print(f, d["value"], d["ms_per_step"], d.get("latency_ms", {}).get("p50"), {k: (v.get("value") if isinstance(v, dict) else v) for k, v in hi.items() if k in ("synchronous", "lookahead")}, d.get("stages_ms", {}).get("solve"))
PY
